// Micro-test for the wrong rows on lanes 48-63, fourth hypothesis: every kernel that showed them has
// SLP-vectorised packed fp32 arithmetic (v_pk_mul_f32 / v_pk_add_f32) that reads, with an op_sel_hi broadcast, a
// register written by the instruction JUST in front of it (the GroupNorm mean / rstd: v_mul_f32, v_div_fixup_f32, or a
// DPP add), with no wait state in between -- hipcc puts none there.  Is the result forwarded correctly on gfx950?
//   A: v_mul_f32 vM, c, vS          ; v_pk_add_f32 vD[0:1], vX[0:1], vM[0:1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]
//   B: v_div_fixup_f32 vM, ...      ; v_pk_mul_f32 vD[0:1], vX[0:1], vM[0:1] op_sel_hi:[1,0]
//   C: v_add_f32_dpp vS, vS, vS row_half_mirror ; v_mul_f32 vM, c, vS ; v_pk_add_f32 ... (the kernels' sequence)
//   D: v_rcp_f32 vM, vS (transcendental) ; v_pk_mul_f32 ... vM broadcast
// each with 0 / 1 / 2 wait states between producer and packed consumer, next to co-resident waves issuing MFMA + LDS.
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/pk_forward.hip -o tools/micro/bin/pk_forward
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define NOPS0 ""
#define NOPS1 "s_nop 0\n\t"
#define NOPS2 "s_nop 1\n\t"

// the producer writes v100; v[100:101] is the packed consumer's operand; v101 holds junk (as in the kernels: whatever
// the allocator left in the high half).  %1 returns what the producer wrote.
#define PRE "v_mov_b32_e32 v101, %4\n\ts_nop 4\n\t"
#define POST "s_nop 4\n\tv_mov_b32_e32 %1, v100\n\ts_nop 1"
#define CASE_A(N) asm volatile(PRE "v_mul_f32_e32 v100, 0x3c000000, %2\n\t" N \
                               "v_pk_add_f32 %0, %3, v[100:101] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t" POST \
                               : "=&v"(d), "=&v"(q) : "v"(s), "v"(x), "v"(junk), "v"(den) : "v100", "v101")
#define CASE_B(N) asm volatile(PRE "v_div_fixup_f32 v100, %2, %5, 1.0\n\t" N \
                               "v_pk_mul_f32 %0, %3, v[100:101] op_sel_hi:[1,0]\n\t" POST \
                               : "=&v"(d), "=&v"(q) : "v"(s), "v"(x), "v"(junk), "v"(den) : "v100", "v101")
#define CASE_C(N) asm volatile(PRE "v_mov_b32_e32 v102, %2\n\ts_nop 1\n\t" \
                               "v_add_f32_dpp v102, v102, v102 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
                               "v_mul_f32_e32 v100, 0x3c000000, v102\n\t" N \
                               "v_pk_add_f32 %0, %3, v[100:101] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t" POST \
                               : "=&v"(d), "=&v"(q) : "v"(s), "v"(x), "v"(junk), "v"(den) : "v100", "v101", "v102")
#define CASE_D(N) asm volatile(PRE "v_rcp_f32_e32 v100, %2\n\t" N \
                               "v_pk_mul_f32 %0, %3, v[100:101] op_sel_hi:[1,0]\n\t" POST \
                               : "=&v"(d), "=&v"(q) : "v"(s), "v"(x), "v"(junk), "v"(den) : "v100", "v101")

__device__ __forceinline__ unsigned lcg(unsigned &s) { s = s * 1664525u + 1013904223u; return s; }
__device__ __forceinline__ float unit(unsigned r) { return __builtin_bit_cast(float, 0x3f800000u | (r >> 9)) - 0.5f; }   // [0.5, 1.5)

template <int CASE, int NOPS>
__global__ __launch_bounds__(512) void k(unsigned *bad_q, unsigned long long *total, float *sink, int iters) {
    __shared__ float lds[2048];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave >= 4) {       // noise: MFMA + LDS traffic on the same SIMDs
        f16x8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(lane * 0.01f + i); b[i] = (_Float16)(i * 0.5f - lane * 0.02f); }
        f32x4 c = {0.f, 0.f, 0.f, 0.f};
        float p = lane;
        for (int it = 0; it < iters; ++it) {
            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
            p = p * 1.0001f + 0.5f;
            lds[(threadIdx.x * 4 + it) & 2047] = p;
            c[0] += lds[(threadIdx.x * 7 + it) & 2047];
        }
        sink[blockIdx.x * 512 + threadIdx.x] = c[0] + c[1] + c[2] + c[3] + p;
        return;
    }
    unsigned sd = 0x9e3779b9u * (blockIdx.x * 512 + threadIdx.x + 1);
    unsigned bad = 0;
    unsigned long long n = 0;
    for (int it = 0; it < iters; ++it) {
        f32x2 x = {unit(lcg(sd)), unit(lcg(sd))}, d = {0.f, 0.f};
        float s = unit(lcg(sd)), den = 1.f + unit(lcg(sd)), junk = unit(lcg(sd)) * 1e6f, q = 0.f;
        if (CASE == 0) { if (NOPS == 0) CASE_A(NOPS0); else if (NOPS == 1) CASE_A(NOPS1); else CASE_A(NOPS2); }
        if (CASE == 1) { if (NOPS == 0) CASE_B(NOPS0); else if (NOPS == 1) CASE_B(NOPS1); else CASE_B(NOPS2); }
        if (CASE == 2) { if (NOPS == 0) CASE_C(NOPS0); else if (NOPS == 1) CASE_C(NOPS1); else CASE_C(NOPS2); }
        if (CASE == 3) { if (NOPS == 0) CASE_D(NOPS0); else if (NOPS == 1) CASE_D(NOPS1); else CASE_D(NOPS2); }
        // q = what the producer wrote, read back well after it (5+ wait states): the packed consumer must have seen it
        const f32x2 want = (CASE == 0 || CASE == 2) ? x - f32x2{q, q} : x * f32x2{q, q};
        if (CASE == 0) bad += q != 0.0078125f * s;
        if (CASE == 2) bad += q != 0.0078125f * (s + __shfl(s, (lane & ~7) | (7 - (lane & 7)), 64));
        bad += (__builtin_bit_cast(unsigned, d.x) != __builtin_bit_cast(unsigned, want.x)) +
               (__builtin_bit_cast(unsigned, d.y) != __builtin_bit_cast(unsigned, want.y));
        n += 2;
    }
    if (bad) atomicAdd(bad_q + (lane >> 4), bad);
    if (lane == 0) atomicAdd(total, n * 64ull);
}

template <int CASE, int NOPS>
void run(int blocks, int iters) {
    unsigned *bad; unsigned long long *tot; float *sink;
    CK(hipMalloc(&bad, 16)); CK(hipMalloc(&tot, 8)); CK(hipMalloc(&sink, (size_t)blocks * 512 * 4));
    CK(hipMemset(bad, 0, 16)); CK(hipMemset(tot, 0, 8));
    hipLaunchKernelGGL((k<CASE, NOPS>), dim3(blocks), dim3(512), 0, 0, bad, tot, sink, iters);
    CK(hipDeviceSynchronize());
    unsigned h[4]; unsigned long long t;
    CK(hipMemcpy(h, bad, 16, hipMemcpyDeviceToHost)); CK(hipMemcpy(&t, tot, 8, hipMemcpyDeviceToHost));
    const char *names[] = {"A v_mul -> v_pk_add (broadcast)", "B v_div_fixup -> v_pk_mul (broadcast)", "C dpp add -> v_mul -> v_pk_add",
                           "D v_rcp (trans) -> v_pk_mul (broadcast)"};
    printf("%-42s wait states %d: %llu results, wrong by lane quarter 0-15 / 16-31 / 32-47 / 48-63: %u %u %u %u\n", names[CASE], NOPS, t,
           h[0], h[1], h[2], h[3]);
    CK(hipFree(bad)); CK(hipFree(tot)); CK(hipFree(sink));
}

int main(int argc, char **argv) {
    const int blocks = argc > 1 ? atoi(argv[1]) : 512, iters = argc > 2 ? atoi(argv[2]) : 20000;
    run<0, 0>(blocks, iters); run<0, 1>(blocks, iters); run<0, 2>(blocks, iters);
    run<1, 0>(blocks, iters); run<1, 1>(blocks, iters); run<1, 2>(blocks, iters);
    run<2, 0>(blocks, iters); run<2, 1>(blocks, iters); run<2, 2>(blocks, iters);
    run<3, 0>(blocks, iters); run<3, 1>(blocks, iters); run<3, 2>(blocks, iters);
    return 0;
}
