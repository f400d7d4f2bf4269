// Micro-benchmark: the MFMA sequence of one LaneConv unit (3 sub-blocks x 2 K-steps x 2 channel blocks x 3 plane
// products = 36 v_mfma_f32_16x16x32_f16 over 8 weight registers-quads, 12 A quads, 6 accumulators), operands in registers.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void k(float *out, unsigned long long *cyc, int iters, const f16x8 *src) {
    f16x8 w[2][2][2], a[3][2][2];      // [plane][ks][cb], [rb][ks][plane]
    for (int i = 0; i < 8; ++i) w[i >> 2][(i >> 1) & 1][i & 1] = src[threadIdx.x + 64 * i];
    for (int i = 0; i < 12; ++i) a[i / 4][(i >> 1) & 1][i & 1] = src[threadIdx.x + 64 * (8 + i)];
    f32x4 c[3][2];
    for (int j = 0; j < 6; ++j) c[j / 2][j & 1] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rb = 0; rb < 3; ++rb)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                if (MODE == 1) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) {
                    f32x4 x = c[rb][cb];
#pragma unroll
                    for (int q = 0; q < 3; ++q) x = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[PB[q]][ks][cb], a[rb][ks][PA[q]], x, 0, 0, 0);
                    c[rb][cb] = x;
                }
            }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int j = 0; j < 6; ++j) s += c[j / 2][j & 1][0] + c[j / 2][j & 1][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int MODE>
void run(int threads, const char *tag) {
    float *out; unsigned long long *cyc; f16x8 *src;
    hipMalloc(&out, threads * 4); hipMalloc(&cyc, (threads / 64) * 8); hipMalloc(&src, 64 * 20 * 16 + 1024 * 16);
    hipMemset(src, 0, 64 * 20 * 16 + 1024 * 16);
    const int iters = 1000;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<MODE>), dim3(1), dim3(threads), 0, 0, out, cyc, iters, src);
    hipDeviceSynchronize();
    unsigned long long h[16];
    hipMemcpy(h, cyc, (threads / 64) * 8, hipMemcpyDeviceToHost);
    printf("%-34s: %.1f cycles per MFMA (wave 0), %.1f (last wave)\n", tag, (double)h[0] / (iters * 36.0), (double)h[threads / 64 - 1] / (iters * 36.0));
}

int main() {
    run<0>(256, "unit pattern, 1 wave/SIMD");
    run<1>(256, "unit pattern + sched barriers");
    run<0>(512, "unit pattern, 2 waves/SIMD");
    return 0;
}
