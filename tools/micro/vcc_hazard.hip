// Micro-test for the round-2 bf16x3 wrong answers: does the sequence hipcc emits for `x < 0 ? 0 : x`
// on gfx950,
//     v_cmp_ngt_f32_e32 vcc, 0, vK ; s_nop 1 ; v_cndmask_b32_e32 vK, 0, vK, vcc          (16 times back to back)
// ever select with a stale VCC?  Each wave runs that exact sequence (explicit wait states NOPS = 0 / 1 / 2 / 4, written
// in inline asm so that the compiler's hazard recognizer adds nothing) on data whose sign pattern changes from one
// compare to the next and from lane to lane, next to co-resident waves that issue MFMAs, packed fp32 VALU and LDS
// traffic (the environment of the failing kernels: 4 waves per SIMD, mixed roles).  Mismatches against an integer
// restatement are counted per lane.
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/vcc_hazard.hip -o gpurun_out/vcc_hazard
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define STR2(x) #x
#define STR(x) STR2(x)
#define STEP(n, r) "v_cmp_ngt_f32_e32 vcc, 0, %" #r "\n\t" n "v_cndmask_b32_e32 %" #r ", 0, %" #r ", vcc\n\t"
#define SEQ(n) STEP(n, 0) STEP(n, 1) STEP(n, 2) STEP(n, 3) STEP(n, 4) STEP(n, 5) STEP(n, 6) STEP(n, 7)

template <int NOPS>
__device__ __forceinline__ void relu8(float (&x)[8]) {
    if (NOPS == 0)
        asm volatile(SEQ("") "s_nop 4" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7])::"vcc");
    else if (NOPS == 1)     // what hipcc emits: 2 wait states
        asm volatile(SEQ("s_nop 1\n\t") "s_nop 4" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7])::"vcc");
    else if (NOPS == 2)
        asm volatile(SEQ("s_nop 2\n\t") "s_nop 4" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7])::"vcc");
    else
        asm volatile(SEQ("s_nop 4\n\t") "s_nop 4" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7])::"vcc");
}

// the same 2-wait-state sequence right behind an exec-masked region (the failing kernels run `if (live) r += res` on a
// subset of lanes just before their ReLU): lanes 16..47 double their values under a partial EXEC, EXEC is restored by
// an SALU write, the compares follow immediately
__device__ __forceinline__ void relu8_after_masked(float (&x)[8]) {
    asm volatile("s_mov_b64 s[20:21], exec\n\t"
                 "s_mov_b32 exec_lo, 0xffff0000\n\ts_mov_b32 exec_hi, 0x0000ffff\n\t"
                 "v_add_f32_e32 %0, %0, %0\n\tv_add_f32_e32 %1, %1, %1\n\tv_add_f32_e32 %2, %2, %2\n\tv_add_f32_e32 %3, %3, %3\n\t"
                 "v_add_f32_e32 %4, %4, %4\n\tv_add_f32_e32 %5, %5, %5\n\tv_add_f32_e32 %6, %6, %6\n\tv_add_f32_e32 %7, %7, %7\n\t"
                 "s_mov_b64 exec, s[20:21]\n\t"
                 SEQ("s_nop 1\n\t") "s_nop 4"
                 : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7])::"vcc", "s20", "s21");
}

// what the scheduler makes of it inside the kernels (csrc/lgcn_att.hip, k_att_pairs_ws, compare + select build): LDS reads
// are issued, and waited for, BETWEEN a compare and its select
//     v_cmp_ngt_f32 vcc, 0, vA ; v_or ; ds_read_b128 ; ds_read_b128 ; v_cndmask vA ... vcc
//     v_cmp_ngt_f32 vcc, 0, vB ; s_waitcnt lgkmcnt(0) ; v_mov ; v_mov ; v_cndmask vB ... vcc
__device__ __forceinline__ void relu8_lds_between(float (&x)[8], unsigned lds_addr) {
    asm volatile(
        "v_cmp_ngt_f32_e32 vcc, 0, %0\n\t"
        "v_or_b32_e32 v110, 0x100, %8\n\t"
        "ds_read_b128 v[100:103], %8\n\t"
        "ds_read_b128 v[104:107], v110\n\t"
        "v_cndmask_b32_e32 %0, 0, %0, vcc\n\t"
        "v_cmp_ngt_f32_e32 vcc, 0, %1\n\t"
        "v_or_b32_e32 v110, 0x200, %8\n\t"
        "v_add_u32_e32 v111, 0x400, %8\n\t"
        "v_cndmask_b32_e32 %1, 0, %1, vcc\n\t"
        "v_cmp_ngt_f32_e32 vcc, 0, %2\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_mov_b32_e32 v108, v101\n\t"
        "v_mov_b32_e32 v109, v102\n\t"
        "v_cndmask_b32_e32 %2, 0, %2, vcc\n\t"
        "v_cmp_ngt_f32_e32 vcc, 0, %3\n\t"
        "ds_read_b128 v[100:103], v110\n\t"
        "v_pk_mul_f32 v[108:109], v[104:105], v[108:109] op_sel:[1,0] op_sel_hi:[0,1]\n\t"
        "v_cndmask_b32_e32 %3, 0, %3, vcc\n\t"
        "v_cmp_ngt_f32_e32 vcc, 0, %4\n\t"
        "ds_read_b128 v[104:107], v111\n\t"
        "s_waitcnt lgkmcnt(1)\n\t"
        "v_cndmask_b32_e32 %4, 0, %4, vcc\n\t"
        "v_cmp_ngt_f32_e32 vcc, 0, %5\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_cndmask_b32_e32 %5, 0, %5, vcc\n\t"
        "v_cmp_ngt_f32_e32 vcc, 0, %6\n\t"
        "ds_write_b64 v110, v[100:101]\n\t"
        "s_nop 0\n\t"
        "v_cndmask_b32_e32 %6, 0, %6, vcc\n\t"
        "v_cmp_ngt_f32_e32 vcc, 0, %7\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_cndmask_b32_e32 %7, 0, %7, vcc\n\t"
        "s_nop 4"
        : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7])
        : "v"(lds_addr)
        : "vcc", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "memory");
}

__device__ __forceinline__ unsigned lcg(unsigned &s) { s = s * 1664525u + 1013904223u; return s; }

// waves 0..3 of a 512-thread workgroup: the ReLU sequence; waves 4..7: MFMA + packed VALU + LDS noise
template <int NOPS>
__global__ __launch_bounds__(512) void k(unsigned *bad_lane, unsigned long long *total, float *sink, int iters) {
    __shared__ float lds[2048];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave >= 4) {
        f16x8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(lane * 0.01f + i); b[i] = (_Float16)(i * 0.5f - lane * 0.02f); }
        f32x4 c = {0.f, 0.f, 0.f, 0.f};
        float2 p = make_float2(lane, wave);
        for (int it = 0; it < iters; ++it) {
            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
            p.x = p.x * 1.0001f + 0.5f; p.y = p.y * 0.9999f - 0.25f;
            lds[(threadIdx.x * 4 + it) & 2047] = p.x;
            c[0] += lds[(threadIdx.x * 7 + it) & 2047];
        }
        sink[blockIdx.x * 512 + threadIdx.x] = c[0] + c[1] + c[2] + c[3] + p.x + p.y;
        return;
    }
    unsigned s = 0x9e3779b9u * (blockIdx.x * 512 + threadIdx.x + 1);
    unsigned bad = 0;
    unsigned long long n = 0;
    for (int it = 0; it < iters; ++it) {
        float x[8], w[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const unsigned r = lcg(s);
            // magnitude in [1, 2), sign from a bit that differs between consecutive registers and between lanes
            const unsigned bits = 0x3f800000u | (r >> 9) | ((r ^ (r >> 7) ^ (unsigned)(lane + j)) << 31);
            x[j] = __builtin_bit_cast(float, bits);
            w[j] = (bits >> 31) ? 0.f : (NOPS == 5 && lane >= 16 && lane < 48 ? 2.f * x[j] : x[j]);
        }
        if (NOPS == 5) relu8_after_masked(x);
        else if (NOPS == 6) relu8_lds_between(x, (unsigned)((threadIdx.x & 255) * 16));
        else relu8<NOPS>(x);
#pragma unroll
        for (int j = 0; j < 8; ++j) bad += __builtin_bit_cast(unsigned, x[j]) != __builtin_bit_cast(unsigned, w[j]);
        n += 8;
    }
    if (bad) atomicAdd(bad_lane + lane, bad);
    if (lane == 0) atomicAdd(total, n * 64ull);
}

template <int NOPS>
void run(int blocks, int iters) {
    unsigned *bad; unsigned long long *tot; float *sink;
    hipMalloc(&bad, 64 * 4); hipMalloc(&tot, 8); hipMalloc(&sink, (size_t)blocks * 512 * 4);
    hipMemset(bad, 0, 64 * 4); hipMemset(tot, 0, 8);
    hipLaunchKernelGGL((k<NOPS>), dim3(blocks), dim3(512), 0, 0, bad, tot, sink, iters);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); exit(1); }
    unsigned h[64]; unsigned long long t;
    hipMemcpy(h, bad, 64 * 4, hipMemcpyDeviceToHost); hipMemcpy(&t, tot, 8, hipMemcpyDeviceToHost);
    unsigned long long nb = 0; unsigned q[4] = {0, 0, 0, 0};
    for (int i = 0; i < 64; ++i) { nb += h[i]; q[i >> 4] += h[i]; }
    if (NOPS == 5) printf("behind an exec-masked region, ");
    if (NOPS == 6) printf("LDS reads issued and waited for between compare and select, ");
    printf("wait states %d (s_nop %s): %llu selects, %llu wrong; by lane quarter 0-15 / 16-31 / 32-47 / 48-63: %u %u %u %u\n",
           NOPS == 0 ? 0 : NOPS == 5 ? 2 : NOPS + 1, NOPS == 0 ? "none" : (NOPS == 1 || NOPS == 5) ? "1 = hipcc's" : NOPS == 2 ? "2" : "4", t, nb, q[0], q[1], q[2], q[3]);
    hipFree(bad); hipFree(tot); hipFree(sink);
}

int main(int argc, char **argv) {
    const int blocks = argc > 1 ? atoi(argv[1]) : 512, iters = argc > 2 ? atoi(argv[2]) : 20000;
    run<1>(blocks, iters);
    run<0>(blocks, iters);
    run<2>(blocks, iters);
    run<4>(blocks, iters);
    run<5>(blocks, iters);
    run<6>(blocks, iters);
    run<1>(blocks, iters);
    return 0;
}
