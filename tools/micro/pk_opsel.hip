// Micro-test for the wrong rows on lanes 48-63: tools/micro/pk_producer.hip showed that the CONSUMER
// v_pk_mul_f32 ... op_sel:[0,1] op_sel_hi:[1,0] (second source's halves swapped) returns wrong results in lanes 48-63
// whatever produced its operands and however long ago.  This test isolates the instruction: operands written >= 5 wait
// states earlier, one instruction under test, results read >= 5 wait states later; per operand-select form, with and
// without co-resident waves that issue MFMA + LDS traffic, and what a wrong result equals.
// Build: hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize tools/micro/pk_opsel.hip -o tools/micro/bin/pk_opsel
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// operands: %0 / %1 results; %2 a (pair), %3 b (pair), %4 c (pair), %5 junk
#define PRE "v_mov_b32_e32 v108, %5\n\tv_mov_b32_e32 v109, %5\n\ts_nop 4\n\t"
#define POST "s_nop 4\n\tv_mov_b32_e32 %0, v108\n\tv_mov_b32_e32 %1, v109"
#define T(INSN) asm volatile(PRE INSN "\n\t" POST : "=&v"(d0), "=&v"(d1) : "v"(a), "v"(b), "v"(c), "v"(junk) : "v108", "v109")

__device__ __forceinline__ unsigned lcg(unsigned &s) { s = s * 1664525u + 1013904223u; return s; }
__device__ __forceinline__ float unit(unsigned r) { return __builtin_bit_cast(float, 0x3f800000u | (r >> 9)) - 1.5f; }   // [-0.5, 0.5)
__device__ __forceinline__ unsigned bits(float v) { return __builtin_bit_cast(unsigned, v); }

constexpr int NCASE = 11;
template <int CASE, bool NOISE>
__global__ __launch_bounds__(512) void k(unsigned *bad_q, unsigned *kind, unsigned long long *total, float *sink, int iters) {
    __shared__ float lds[2048];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave >= 4) {
        if (!NOISE) return;
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
    unsigned bad = 0, k_unswapped = 0, k_junk = 0, k_lo = 0, k_hi = 0;
    unsigned long long n = 0;
    for (int it = 0; it < iters; ++it) {
        const float ax = unit(lcg(sd)), ay = unit(lcg(sd)), bx = unit(lcg(sd)), by = unit(lcg(sd)), cx = unit(lcg(sd)), cy = unit(lcg(sd));
        const float junk = 1e6f + unit(lcg(sd));
        const f32x2 a = {ax, ay}, b = {bx, by}, c = {cx, cy};
        float d0 = 0.f, d1 = 0.f, w0, w1, u0, u1;      // w: expected; u: what the plain (no operand select) form would give
        if (CASE == 0) { T("v_pk_mul_f32 v[108:109], %2, %3 op_sel:[0,1] op_sel_hi:[1,0]"); w0 = __fmul_rn(ax, by); w1 = __fmul_rn(ay, bx); u0 = __fmul_rn(ax, bx); u1 = __fmul_rn(ay, by); }
        if (CASE == 1) { T("v_pk_add_f32 v[108:109], %2, %3 op_sel:[0,1] op_sel_hi:[1,0]"); w0 = __fadd_rn(ax, by); w1 = __fadd_rn(ay, bx); u0 = __fadd_rn(ax, bx); u1 = __fadd_rn(ay, by); }
        if (CASE == 2) { T("v_pk_add_f32 v[108:109], %2, %2 op_sel:[0,1] op_sel_hi:[1,0]"); w0 = __fadd_rn(ax, ay); w1 = __fadd_rn(ay, ax); u0 = __fadd_rn(ax, ax); u1 = __fadd_rn(ay, ay); }
        if (CASE == 3) { T("v_pk_mul_f32 v[108:109], %2, %3"); w0 = __fmul_rn(ax, bx); w1 = __fmul_rn(ay, by); u0 = w0; u1 = w1; }
        if (CASE == 4) { T("v_pk_mul_f32 v[108:109], %2, %3 op_sel_hi:[1,0]"); w0 = __fmul_rn(ax, bx); w1 = __fmul_rn(ay, bx); u0 = w0; u1 = __fmul_rn(ay, by); }
        if (CASE == 5) { T("v_pk_mul_f32 v[108:109], %2, %3 op_sel:[1,0] op_sel_hi:[0,1]"); w0 = __fmul_rn(ay, bx); w1 = __fmul_rn(ax, by); u0 = __fmul_rn(ax, bx); u1 = __fmul_rn(ay, by); }
        if (CASE == 6) { T("v_pk_fma_f32 v[108:109], %2, %3, %4 op_sel:[1,0,0]"); w0 = __fmaf_rn(ay, bx, cx); w1 = __fmaf_rn(ay, by, cy); u0 = __fmaf_rn(ax, bx, cx); u1 = w1; }
        if (CASE == 7) { T("v_pk_mov_b32 v[108:109], %2, %3 op_sel:[1,0]"); w0 = ay; w1 = bx; u0 = ax; u1 = by; }
        if (CASE == 8) { T("v_pk_add_f32 v[108:109], %2, %3 op_sel_hi:[0,1]"); w0 = __fadd_rn(ax, bx); w1 = __fadd_rn(ax, by); u0 = w0; u1 = __fadd_rn(ay, by); }
        if (CASE == 9) { T("v_pk_mul_f32 v[108:109], %2, %3 op_sel:[0,1]"); w0 = __fmul_rn(ax, by); w1 = __fmul_rn(ay, by); u0 = __fmul_rn(ax, bx); u1 = w1; }
        if (CASE == 10) { T("v_pk_add_f32 v[108:109], %2, %3 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]"); w0 = __fsub_rn(ax, bx); w1 = __fsub_rn(ay, bx); u0 = w0; u1 = __fsub_rn(ay, by); }
        const bool e0 = bits(d0) != bits(w0), e1 = bits(d1) != bits(w1);
        bad += e0 + e1;
        k_lo += e0; k_hi += e1;
        k_unswapped += (e0 && bits(d0) == bits(u0)) + (e1 && bits(d1) == bits(u1));
        k_junk += (e0 && bits(d0) == bits(junk)) + (e1 && bits(d1) == bits(junk));
        n += 2;
    }
    if (bad) {
        atomicAdd(bad_q + (lane >> 4), bad);
        atomicAdd(kind + 0, k_lo); atomicAdd(kind + 1, k_hi); atomicAdd(kind + 2, k_unswapped); atomicAdd(kind + 3, k_junk);
    }
    if (lane == 0) atomicAdd(total, n * 64ull);
}

template <int CASE, bool NOISE>
void run(int blocks, int iters) {
    unsigned *bad; unsigned long long *tot; float *sink;
    CK(hipMalloc(&bad, 32)); CK(hipMalloc(&tot, 8)); CK(hipMalloc(&sink, (size_t)blocks * 512 * 4));
    CK(hipMemset(bad, 0, 32)); CK(hipMemset(tot, 0, 8));
    hipLaunchKernelGGL((k<CASE, NOISE>), dim3(blocks), dim3(512), 0, 0, bad, bad + 4, tot, sink, iters);
    CK(hipDeviceSynchronize());
    unsigned h[8]; unsigned long long t;
    CK(hipMemcpy(h, bad, 32, hipMemcpyDeviceToHost)); CK(hipMemcpy(&t, tot, 8, hipMemcpyDeviceToHost));
    const char *names[NCASE] = {"v_pk_mul_f32 a, b op_sel:[0,1] op_sel_hi:[1,0]", "v_pk_add_f32 a, b op_sel:[0,1] op_sel_hi:[1,0]",
                                "v_pk_add_f32 a, a op_sel:[0,1] op_sel_hi:[1,0]", "v_pk_mul_f32 a, b", "v_pk_mul_f32 a, b op_sel_hi:[1,0]",
                                "v_pk_mul_f32 a, b op_sel:[1,0] op_sel_hi:[0,1]", "v_pk_fma_f32 a, b, c op_sel:[1,0,0]", "v_pk_mov_b32 a, b op_sel:[1,0]",
                                "v_pk_add_f32 a, b op_sel_hi:[0,1]", "v_pk_mul_f32 a, b op_sel:[0,1]", "v_pk_add_f32 a, -b op_sel_hi:[1,0]"};
    printf("%-48s %-9s %llu results, wrong by lane quarter: %u %u %u %u; wrong low / high half: %u / %u; equal to the unselected form: %u, to the "
           "destination's old value: %u\n", names[CASE], NOISE ? "MFMA+LDS" : "alone", t, h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
    CK(hipFree(bad)); CK(hipFree(tot)); CK(hipFree(sink));
}

template <int CASE>
void both(int blocks, int iters) { run<CASE, true>(blocks, iters); run<CASE, false>(blocks, iters); }

int main(int argc, char **argv) {
    const int blocks = argc > 1 ? atoi(argv[1]) : 512, iters = argc > 2 ? atoi(argv[2]) : 10000;
    both<0>(blocks, iters); both<1>(blocks, iters); both<2>(blocks, iters); both<3>(blocks, iters); both<4>(blocks, iters); both<5>(blocks, iters);
    both<6>(blocks, iters); both<7>(blocks, iters); both<8>(blocks, iters); both<9>(blocks, iters); both<10>(blocks, iters);
    return 0;
}
