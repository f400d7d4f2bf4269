// Micro-test for the round-2 bf16x3 wrong answers, second hypothesis: the three kernels that failed are
// exactly the three that spill registers to SCRATCH (private segment 20 / 20 / 44 bytes per lane); the same sources with
// no spill never failed.  Does a wave ever read back from its private segment something it did not write, when kernels
// with DIFFERENT private-segment sizes run concurrently on several HIP streams (several hardware queues), eagerly and
// from captured graphs -- the way the forward runs them?
// Every thread keeps WORDS dwords in a volatile local array (forced to scratch), fills them with a pattern of
// (launch id, global thread id, word), spends a data-dependent time in LDS / barrier work, and checks them; mismatches are
// counted per lane quarter.
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/scratch_race.hip -o tools/micro/bin/scratch_race
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int WORDS>
__global__ __launch_bounds__(512) void k(unsigned *bad_q, unsigned long long *checked, unsigned launch, int spin) {
    __shared__ float lds[1024];
    volatile unsigned priv[WORDS];                 // volatile: lives in the private segment (scratch_store / scratch_load)
    const unsigned gid = blockIdx.x * 512u + threadIdx.x;
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int w = 0; w < WORDS; ++w) priv[w] = (launch * 0x9e3779b9u) ^ (gid * 2654435761u) ^ (unsigned)(w * 0x85ebca6bu + WORDS);
    float acc = gid * 1e-6f;
    const int n = spin + (int)((gid * 7u + launch) % 97u);     // waves of a workgroup and workgroups drift apart
    for (int i = 0; i < n; ++i) {
        lds[(threadIdx.x + i) & 1023] = acc;
        acc = acc * 1.0001f + lds[(threadIdx.x * 5 + i) & 1023];
        if ((i & 31) == 31) __syncthreads();
    }
    unsigned bad = 0;
#pragma unroll
    for (int w = 0; w < WORDS; ++w)
        bad += priv[w] != ((launch * 0x9e3779b9u) ^ (gid * 2654435761u) ^ (unsigned)(w * 0x85ebca6bu + WORDS));
    if (bad) atomicAdd(bad_q + (lane >> 4), bad);
    if (threadIdx.x == 0) atomicAdd(checked, 512ull * WORDS);
    if (acc == 123.456f) lds[0] = acc;             // keeps the spin alive
}

int main(int argc, char **argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 300, blocks = argc > 2 ? atoi(argv[2]) : 324;
    unsigned *bad; unsigned long long *chk;
    CK(hipMalloc(&bad, 16)); CK(hipMalloc(&chk, 8));
    CK(hipMemset(bad, 0, 16)); CK(hipMemset(chk, 0, 8));
    hipStream_t st[4];
    for (auto &s : st) CK(hipStreamCreate(&s));
    unsigned launch = 1;
    auto burst = [&](hipStream_t s, int j) {       // the mix of one forward: private segments of 5, 11 and 0 dwords, tiny and chip-filling grids
        hipLaunchKernelGGL((k<5>), dim3(blocks), dim3(512), 0, s, bad, chk, launch++, 200 + 50 * j);
        hipLaunchKernelGGL((k<11>), dim3(50 + 37 * j), dim3(512), 0, s, bad, chk, launch++, 100);
        hipLaunchKernelGGL((k<5>), dim3(100), dim3(512), 0, s, bad, chk, launch++, 300);
        hipLaunchKernelGGL((k<32>), dim3(blocks), dim3(512), 0, s, bad, chk, launch++, 50);
    };
    // (1) eager, four streams
    for (int r = 0; r < rounds; ++r)
        for (int j = 0; j < 4; ++j) burst(st[j], j);
    CK(hipDeviceSynchronize());
    unsigned h[4]; unsigned long long c;
    CK(hipMemcpy(h, bad, 16, hipMemcpyDeviceToHost)); CK(hipMemcpy(&c, chk, 8, hipMemcpyDeviceToHost));
    printf("eager, 4 streams: %llu private dwords checked, wrong by lane quarter 0-15 / 16-31 / 32-47 / 48-63: %u %u %u %u\n", c, h[0], h[1], h[2], h[3]);
    // (2) four captured graphs replayed concurrently
    CK(hipMemset(bad, 0, 16)); CK(hipMemset(chk, 0, 8));
    hipGraphExec_t ex[4];
    for (int j = 0; j < 4; ++j) {
        hipGraph_t g;
        CK(hipStreamBeginCapture(st[j], hipStreamCaptureModeThreadLocal));
        for (int q = 0; q < 6; ++q) burst(st[j], j);
        CK(hipStreamEndCapture(st[j], &g));
        CK(hipGraphInstantiate(&ex[j], g, nullptr, nullptr, 0));
    }
    for (int r = 0; r < rounds / 3 + 1; ++r)
        for (int j = 0; j < 4; ++j) CK(hipGraphLaunch(ex[j], st[j]));
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h, bad, 16, hipMemcpyDeviceToHost)); CK(hipMemcpy(&c, chk, 8, hipMemcpyDeviceToHost));
    printf("4 graphs in flight: %llu private dwords checked, wrong by lane quarter: %u %u %u %u\n", c, h[0], h[1], h[2], h[3]);
    return 0;
}
