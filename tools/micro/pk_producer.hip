// Micro-test for the wrong rows on lanes 48-63, fifth hypothesis: a PACKED instruction as the PRODUCER.
// The compare + select ReLU build differs from the shipped one in that hipcc's SLP vectoriser turns the GroupNorm row
// arithmetic into v_pk_add_f32 / v_pk_mul_f32 / v_pk_mov_b32 chains whose results are consumed a few instructions later.
// Which producer -> consumer distances are safe on gfx950, in every lane quarter?
//   producers: P1 v_pk_mov_b32 v[100:101], x, y op_sel:[1,0]      P2 v_pk_mul_f32 v[100:101], x, y      P3 two v_mov_b32
//   consumers: C1 v_pk_mul_f32 .., v[100:101], y op_sel:[0,1] op_sel_hi:[1,0]   C2 v_pk_mul_f32 .., v[100:101], y
//              C3 two v_mov_b32                                                  C4 v_pk_add_f32 .., v[100:101], v[100:101] op_sel_hi:[0,1]
//   gaps: 0 .. 6 wait states (s_nop), 1 / 2 / 3 independent v_mov_b32, 1 independent v_pk_mul_f32
// v[100:101] hold junk before the producer, so a consumer that reads too early sees the junk.  Waves 4-7 of every
// workgroup issue MFMA + LDS traffic on the same SIMDs.
// Build: hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize tools/micro/pk_producer.hip -o tools/micro/bin/pk_producer
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// operands: %0 / %1 results; %2 x (pair), %3 y (pair), %4 x.y, %5 y.x, %6 junk
#define PRE "v_mov_b32_e32 v100, %6\n\tv_mov_b32_e32 v101, %6\n\tv_mov_b32_e32 v106, %6\n\tv_mov_b32_e32 v107, %6\n\ts_nop 4\n\t"
#define P1 "v_pk_mov_b32 v[100:101], %2, %3 op_sel:[1,0]\n\t"
#define P2 "v_pk_mul_f32 v[100:101], %2, %3\n\t"
#define P3 "v_mov_b32_e32 v100, %4\n\tv_mov_b32_e32 v101, %5\n\t"
#define C1 "v_pk_mul_f32 v[108:109], v[100:101], %3 op_sel:[0,1] op_sel_hi:[1,0]\n\t"
#define C2 "v_pk_mul_f32 v[108:109], v[100:101], %3\n\t"
#define C3 "v_mov_b32_e32 v108, v100\n\tv_mov_b32_e32 v109, v101\n\t"
#define C4 "v_pk_add_f32 v[108:109], v[100:101], v[100:101] op_sel_hi:[0,1]\n\t"
#define POST "s_nop 4\n\tv_mov_b32_e32 %0, v108\n\tv_mov_b32_e32 %1, v109"
#define G0 ""
#define G1 "s_nop 0\n\t"
#define G2 "s_nop 1\n\t"
#define G3 "s_nop 2\n\t"
#define G4 "s_nop 3\n\t"
#define G5 "s_nop 4\n\t"
#define G6 "s_nop 5\n\t"
#define G7 "v_mov_b32_e32 v104, v106\n\t"
#define G8 "v_mov_b32_e32 v104, v106\n\tv_mov_b32_e32 v105, v107\n\t"
#define G9 "v_mov_b32_e32 v104, v106\n\tv_mov_b32_e32 v105, v107\n\tv_mov_b32_e32 v103, v106\n\t"
#define G10 "v_pk_mul_f32 v[104:105], v[106:107], v[106:107]\n\t"
#define CLOB "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109"
#define RUN(P, G, C) asm volatile(PRE P G C POST : "=&v"(d0), "=&v"(d1) : "v"(x), "v"(y), "v"(xy), "v"(yx), "v"(junk) : CLOB)
#define GAPS(P, C) do { switch (GAP) { case 0: RUN(P, G0, C); break; case 1: RUN(P, G1, C); break; case 2: RUN(P, G2, C); break; \
    case 3: RUN(P, G3, C); break; case 4: RUN(P, G4, C); break; case 5: RUN(P, G5, C); break; case 6: RUN(P, G6, C); break;      \
    case 7: RUN(P, G7, C); break; case 8: RUN(P, G8, C); break; case 9: RUN(P, G9, C); break; default: RUN(P, G10, C); break; } } while (0)
#define CONSUMERS(P) do { if (CONS == 1) GAPS(P, C1); else if (CONS == 2) GAPS(P, C2); else if (CONS == 3) GAPS(P, C3); else GAPS(P, C4); } while (0)

__device__ __forceinline__ unsigned lcg(unsigned &s) { s = s * 1664525u + 1013904223u; return s; }
__device__ __forceinline__ float unit(unsigned r) { return __builtin_bit_cast(float, 0x3f800000u | (r >> 9)) - 1.5f; }   // [-0.5, 0.5)
__device__ __forceinline__ unsigned bits(float v) { return __builtin_bit_cast(unsigned, v); }

template <int PROD, int CONS, int GAP>
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
        const float xx = unit(lcg(sd)), xy = unit(lcg(sd)), yx = unit(lcg(sd)), yy = unit(lcg(sd)), junk = 1e6f + unit(lcg(sd));
        const f32x2 x = {xx, xy}, y = {yx, yy};
        float d0 = 0.f, d1 = 0.f;
        if (PROD == 1) CONSUMERS(P1); else if (PROD == 2) CONSUMERS(P2); else CONSUMERS(P3);
        // what the producer leaves in v100 / v101
        const float a = PROD == 2 ? __fmul_rn(xx, yx) : xy, b = PROD == 2 ? __fmul_rn(xy, yy) : yx;
        float w0, w1;
        if (CONS == 1) { w0 = __fmul_rn(a, yy); w1 = __fmul_rn(b, yx); }
        else if (CONS == 2) { w0 = __fmul_rn(a, yx); w1 = __fmul_rn(b, yy); }
        else if (CONS == 3) { w0 = a; w1 = b; }
        else { w0 = __fadd_rn(a, a); w1 = __fadd_rn(a, b); }
        bad += (bits(d0) != bits(w0)) + (bits(d1) != bits(w1));
        n += 2;
    }
    if (bad) atomicAdd(bad_q + (lane >> 4), bad);
    if (lane == 0) atomicAdd(total, n * 64ull);
}

template <int PROD, int CONS, int GAP>
void run(int blocks, int iters) {
    unsigned *bad; unsigned long long *tot; float *sink;
    CK(hipMalloc(&bad, 16)); CK(hipMalloc(&tot, 8)); CK(hipMalloc(&sink, (size_t)blocks * 512 * 4));
    CK(hipMemset(bad, 0, 16)); CK(hipMemset(tot, 0, 8));
    hipLaunchKernelGGL((k<PROD, CONS, GAP>), dim3(blocks), dim3(512), 0, 0, bad, tot, sink, iters);
    CK(hipDeviceSynchronize());
    unsigned h[4]; unsigned long long t;
    CK(hipMemcpy(h, bad, 16, hipMemcpyDeviceToHost)); CK(hipMemcpy(&t, tot, 8, hipMemcpyDeviceToHost));
    const char *prods[] = {"", "v_pk_mov_b32 op_sel:[1,0]", "v_pk_mul_f32", "2 x v_mov_b32"};
    const char *cons[] = {"", "v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[1,0]", "v_pk_mul_f32", "2 x v_mov_b32", "v_pk_add_f32 op_sel_hi:[0,1]"};
    const char *gaps[] = {"0 wait states", "s_nop 0", "s_nop 1", "s_nop 2", "s_nop 3", "s_nop 4", "s_nop 5", "1 independent v_mov", "2 independent v_mov",
                          "3 independent v_mov", "1 independent v_pk_mul"};
    printf("%-26s -> %-42s %-22s %llu results, wrong by lane quarter 0-15 / 16-31 / 32-47 / 48-63: %u %u %u %u\n", prods[PROD], cons[CONS],
           gaps[GAP], t, h[0], h[1], h[2], h[3]);
    CK(hipFree(bad)); CK(hipFree(tot)); CK(hipFree(sink));
}

template <int PROD, int CONS>
void run_gaps(int blocks, int iters) {
    run<PROD, CONS, 0>(blocks, iters); run<PROD, CONS, 1>(blocks, iters); run<PROD, CONS, 2>(blocks, iters); run<PROD, CONS, 3>(blocks, iters);
    run<PROD, CONS, 4>(blocks, iters); run<PROD, CONS, 5>(blocks, iters); run<PROD, CONS, 6>(blocks, iters); run<PROD, CONS, 7>(blocks, iters);
    run<PROD, CONS, 8>(blocks, iters); run<PROD, CONS, 9>(blocks, iters); run<PROD, CONS, 10>(blocks, iters);
}

template <int PROD>
void run_prod(int blocks, int iters) {
    run_gaps<PROD, 1>(blocks, iters); run_gaps<PROD, 2>(blocks, iters); run_gaps<PROD, 3>(blocks, iters); run_gaps<PROD, 4>(blocks, iters);
}

int main(int argc, char **argv) {
    const int blocks = argc > 1 ? atoi(argv[1]) : 512, iters = argc > 2 ? atoi(argv[2]) : 10000;
    run_prod<1>(blocks, iters); run_prod<2>(blocks, iters); run_prod<3>(blocks, iters);
    return 0;
}
