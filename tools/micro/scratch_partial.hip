// Micro-test for the wrong rows on lanes 48-63, the spill pattern of k_att_pairs_ws in the compare +
// select ReLU build, with the REAL instructions (inline asm; hipcc's own volatile arrays become flat_ accesses, not
// scratch_ ones): an 8-byte private slot written at full EXEC, overwritten for a lane subset, the workgroup barrier,
// reloaded under a SPARSE EXEC mask (lanes 0, 8, ..., 56 -- the lanes that publish a tile's pair indices) and compared
// with a shadow copy in registers; 16-byte slots the same way.
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/scratch_partial.hip -o tools/micro/bin/scratch_partial
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned mix(unsigned a, unsigned b) { a ^= b * 0x9e3779b9u; a ^= a >> 15; a *= 0x85ebca6bu; a ^= a >> 13; return a; }
__device__ __forceinline__ void st2(u32x2 v) { asm volatile("scratch_store_dwordx2 off, %0, off" ::"v"(v) : "memory"); }
__device__ __forceinline__ void st4(u32x4 v) { asm volatile("scratch_store_dwordx4 off, %0, off offset:16" ::"v"(v) : "memory"); }
__device__ __forceinline__ u32x2 ld2() { u32x2 v; asm volatile("scratch_load_dwordx2 %0, off, off\n\ts_waitcnt vmcnt(0)" : "=v"(v)::"memory"); return v; }
__device__ __forceinline__ u32x4 ld4() { u32x4 v; asm volatile("scratch_load_dwordx4 %0, off, off offset:16\n\ts_waitcnt vmcnt(0)" : "=v"(v)::"memory"); return v; }

__global__ __launch_bounds__(512) void k(unsigned *bad_q, unsigned long long *checked, int iters, unsigned seed) {
    __shared__ float lds[2048];
    volatile unsigned frame[16];       // 64 bytes of private segment at offset 0: what the asm above addresses
    frame[seed & 15] = seed;            // dynamic index: the whole array stays a stack object
    const unsigned gid = blockIdx.x * 512u + threadIdx.x;
    const int lane = threadIdx.x & 63;
    u32x2 h2 = {mix(gid, 1u), mix(gid, 2u)};
    u32x4 h4 = {mix(gid, 3u), mix(gid, 4u), mix(gid, 5u), mix(gid, 6u)};
    st2(h2); st4(h4);
    unsigned bad = 0, n = 0;
    float acc = gid * 1e-6f;
    for (int it = 0; it < iters; ++it) {
        const unsigned r = mix(seed + it, blockIdx.x);               // wave-uniform
        // the kernel's sequence: default value at full EXEC, then the loaded value for the lanes whose next pair exists
        h2 = u32x2{0u, h2.y}; st2(h2);
        if (mix(r, lane >> (r & 3)) & 1) { h2 = u32x2{mix(h2.y, it), mix(gid, it)}; asm volatile("s_waitcnt vmcnt(0)"); st2(h2); }
        if (mix(r, 77u + (lane >> ((r >> 2) & 3))) & 1) { h4 = u32x4{mix(h4.x, it), mix(h4.y, gid), mix(h4.z, 3u), mix(h4.w, 5u)}; st4(h4); }
        for (int q = 0; q < (int)(r & 7); ++q) { lds[(threadIdx.x + q) & 2047] = acc; acc = acc * 1.0001f + lds[(threadIdx.x * 3 + q) & 2047]; }
        __builtin_amdgcn_s_barrier();
        if ((lane & 7) == 0) { const u32x2 v = ld2(); bad += (v.x != h2.x) + (v.y != h2.y); n += 2; }        // sparse mask
        if (mix(r, 5u + lane) & 1) { const u32x4 v = ld4(); bad += (v.x != h4.x) + (v.y != h4.y) + (v.z != h4.z) + (v.w != h4.w); n += 4; }
        if ((r & 0x30) == 0 && (lane & 7) == 0) { h2.y ^= it; st2(h2); }                                        // re-spill under the sparse mask
    }
    { const u32x2 v = ld2(); bad += (v.x != h2.x) + (v.y != h2.y); n += 2; }
    { const u32x4 v = ld4(); bad += (v.x != h4.x) + (v.y != h4.y) + (v.z != h4.z) + (v.w != h4.w); n += 4; }
    if (bad) atomicAdd(bad_q + (lane >> 4), bad);
    atomicAdd(checked, (unsigned long long)n);
    if (acc == 123.456f) lds[0] = acc + frame[(seed >> 4) & 15];
}

int main(int argc, char **argv) {
    const int launches = argc > 1 ? atoi(argv[1]) : 200, blocks = argc > 2 ? atoi(argv[2]) : 512, iters = argc > 3 ? atoi(argv[3]) : 400;
    unsigned *bad; unsigned long long *chk;
    CK(hipMalloc(&bad, 16)); CK(hipMalloc(&chk, 8));
    CK(hipMemset(bad, 0, 16)); CK(hipMemset(chk, 0, 8));
    hipStream_t st[4];
    for (auto &s : st) CK(hipStreamCreate(&s));
    for (int l = 0; l < launches; ++l)
        hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, st[l & 3], bad, chk, iters, 1000u * l);
    CK(hipDeviceSynchronize());
    unsigned h[4]; unsigned long long c;
    CK(hipMemcpy(h, bad, 16, hipMemcpyDeviceToHost)); CK(hipMemcpy(&c, chk, 8, hipMemcpyDeviceToHost));
    printf("scratch_store/load_dwordx2/x4 (off, off), full-EXEC store + partial overwrite + barrier + sparse-mask reload: %llu dwords "
           "checked, wrong by lane quarter 0-15 / 16-31 / 32-47 / 48-63: %u %u %u %u\n", c, h[0], h[1], h[2], h[3]);
    return 0;
}
