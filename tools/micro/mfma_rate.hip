// Micro-benchmark: issue rate of v_mfma_f32_16x16x32_f16 on one CU (gfx950), 1 or 2 waves per SIMD, 1..6 accumulation
// chains, operands in registers.  Build: hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_rate.hip -o gpurun_out/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int CH>
__global__ void k(float *out, unsigned long long *cyc, int iters) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f - threadIdx.x * 0.002f); }
    f32x4 c[CH];
    for (int j = 0; j < CH; ++j) c[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 12 / CH; ++r)
#pragma unroll
            for (int j = 0; j < CH; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c[j], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int j = 0; j < CH; ++j) s += c[j][0] + c[j][1] + c[j][2] + c[j][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int CH>
void run(int threads, int blocks, const char *tag) {
    float *out; unsigned long long *cyc;
    hipMalloc(&out, blocks * threads * 4); hipMalloc(&cyc, blocks * (threads / 64) * 8);
    const int iters = 2000;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<CH>), dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    unsigned long long h[64];
    hipMemcpy(h, cyc, (threads / 64) * 8, hipMemcpyDeviceToHost);
    printf("%-28s chains %d: %.1f cycles per MFMA per wave (wave 0), %.1f (last wave)\n", tag, CH, (double)h[0] / (iters * 12.0),
           (double)h[threads / 64 - 1] / (iters * 12.0));
    hipFree(out); hipFree(cyc);
}

int main() {
    run<1>(256, 1, "1 wave/SIMD, 1 block"); run<2>(256, 1, "1 wave/SIMD, 1 block"); run<4>(256, 1, "1 wave/SIMD, 1 block"); run<6>(256, 1, "1 wave/SIMD, 1 block");
    run<2>(512, 1, "2 waves/SIMD, 1 block"); run<6>(512, 1, "2 waves/SIMD, 1 block");
    run<2>(512, 256, "2 waves/SIMD, 256 blocks"); run<6>(512, 256, "2 waves/SIMD, 256 blocks");
    return 0;
}
