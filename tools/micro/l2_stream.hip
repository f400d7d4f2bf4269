// Micro-benchmark: the rate at which the CUs pull an L2-resident 1 MB table (the LaneConv weight set of one layer)
// when EVERY workgroup streams the whole table -- the access pattern of k_lc_tile's weight slices -- by waves per
// workgroup, workgroups per CU, loads in flight per wave, and whether all workgroups walk the table in the same order
// or each starts at its own 64 KB slice (the unit rotation of k_lc_plan).
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/l2_stream.hip -o tools/micro/bin/l2_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// table: n_slices slices of 64 KB; a workgroup reads slice (s + rot * blockIdx) % n_slices for s = 0 .. n_slices - 1,
// wave w of W reading the w-th 1/W of the slice, 1 KB per wave-instruction, DEPTH instructions in flight.
template <int DEPTH>
__global__ __launch_bounds__(1024) void k(const u32x4 *__restrict__ tab, int n_slices, int rot, int passes, unsigned *sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, W = blockDim.x >> 6;
    const int per_wave = 4096 / W;                       // uint4 words of a 64 KB slice per wave
    u32x4 acc = {0u, 0u, 0u, 0u};
    for (int ps = 0; ps < passes; ++ps)
        for (int s = 0; s < n_slices; ++s) {
            const int sl = (s + rot * (int)blockIdx.x) % n_slices;
            const u32x4 *q = tab + (size_t)sl * 4096 + wave * per_wave + lane;
            for (int i = 0; i < per_wave; i += 64 * DEPTH) {
                u32x4 v[DEPTH];
#pragma unroll
                for (int d = 0; d < DEPTH; ++d) v[d] = q[i + 64 * d];
#pragma unroll
                for (int d = 0; d < DEPTH; ++d) acc ^= v[d];
            }
        }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}

template <int DEPTH>
static void run(const u32x4 *tab, unsigned *sink, int cus, int wg_per_cu, int waves, int rot, int n_slices) {
    const int passes = 8;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const dim3 grid(cus * wg_per_cu), block(64 * waves);
    hipLaunchKernelGGL(k<DEPTH>, grid, block, 0, 0, tab, n_slices, rot, 1, sink);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k<DEPTH>, grid, block, 0, 0, tab, n_slices, rot, passes, sink);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = (double)grid.x * passes * n_slices * 65536.0;
    printf("waves %2d  wg/CU %d  depth %2d  %s  %7.1f us  %6.1f GB/s per CU  %5.2f TB/s chip  (%.1f B/clk/CU at 2.4 GHz)\n", waves, wg_per_cu,
           DEPTH, rot ? "rotated" : "same   ", ms * 1e3, bytes / (ms * 1e-3) / cus / 1e9, bytes / (ms * 1e-3) / 1e12,
           bytes / (ms * 1e-3) / cus / 2.4e9);
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount, n_slices = 16;
    u32x4 *tab; unsigned *sink;
    CK(hipMalloc(&tab, (size_t)n_slices * 65536)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(tab, 1, (size_t)n_slices * 65536)); CK(hipMemset(sink, 0, 4));
    printf("%d CUs, table %d x 64 KB, every workgroup reads all of it\n", cus, n_slices);
    for (int rot = 0; rot < 2; ++rot)
        for (int wg = 1; wg <= 2; ++wg)
            for (int waves : {4, 8, 16}) {
                if (waves * wg > 32) continue;
                run<2>(tab, sink, cus, wg, waves, rot, n_slices);
                run<4>(tab, sink, cus, wg, waves, rot, n_slices);
                if (waves <= 8) run<8>(tab, sink, cus, wg, waves, rot, n_slices);        // 64 * DEPTH words <= a wave's part of a slice
                if (waves <= 4) run<16>(tab, sink, cus, wg, waves, rot, n_slices);
            }
    return 0;
}
