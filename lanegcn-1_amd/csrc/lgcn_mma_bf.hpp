// Split-precision MFMA building blocks shared by lgcn_rowmlp_bf.hip and lgcn_laneconv.hip (gfx950 only):
// operand formats (16-bit planes of an fp32 value), packed-weight fragment loads, the K = 128 pass over an
// LDS-resident A tile, accumulator <-> LDS moves.
#pragma once
#include "lgcn_common.hpp"
#include "lgcn_tile.hpp"

namespace lgcn {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kLDB = kC + 8;  // 16-bit elements per LDS plane row (272 B)

// Operand formats of the split-precision modes.  A value is stored as NP 16-bit planes
// (x = p0 + p1 (+ p2), each plane the rounding of the residual left by the previous ones);
// PROD lists the plane pairs (A plane, B plane) that are multiplied, smallest terms first.
//   F = 0  LGCN_MMA_BF16X3: 3 bf16 planes (3 x 8 bits), 6 products, dropped terms <= 2^-24
//   F = 1  LGCN_MMA_F16X2 : 2 fp16 planes (2 x 11 bits), 3 products, dropped terms <= 2^-22
//          (operands must stay below fp16's 65504: true behind the GroupNorms of this network)
//   F = 2  LGCN_MMA_BF16  : 1 bf16 plane, 1 product
template <int F> struct Fmt;
template <> struct Fmt<0> {
    static constexpr int NP = 3, NPROD = 6;
    static constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
    static __device__ __forceinline__ uint32_t pack(float a, float b) {
        const f32x2 v = {a, b};
        return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));   // v_cvt_pk_bf16_f32 (RNE)
    }
    static __device__ __forceinline__ f32x2 unpack(uint32_t u) {
        return f32x2{__builtin_bit_cast(float, u << 16), __builtin_bit_cast(float, u & 0xffff0000u)};
    }
    static __device__ __forceinline__ f32x4 mfma(uint4 a, uint4 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
};
template <> struct Fmt<1> {
    static constexpr int NP = 2, NPROD = 3;
    static constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};
    static __device__ __forceinline__ uint32_t pack(float a, float b) {
        const f32x2 v = {a, b};
        return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2));    // v_cvt_pk_f16_f32 (RNE)
    }
    static __device__ __forceinline__ f32x2 unpack(uint32_t u) {
        return __builtin_convertvector(__builtin_bit_cast(f16x2, u), f32x2);
    }
    static __device__ __forceinline__ f32x4 mfma(uint4 a, uint4 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
};
template <> struct Fmt<2> : Fmt<0> {
    static constexpr int NP = 1, NPROD = 1;
    static constexpr int PA[1] = {0}, PB[1] = {0};
};

// Diagnostic build only (-DLGCN_STAMPS, tools/stamps.py): s_memtime stamps of the LaneConv phases go to
// the buffer passed as out_pre (never to an output); the shipped library contains no stamp.
#ifdef LGCN_STAMPS
__device__ __forceinline__ unsigned long long stamp() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define LGCN_STAMP(slot) do { if (lane == 0 && sbuf && (slot) < 64) sbuf[(slot)] = stamp(); } while (0)
#else
#define LGCN_STAMP(slot) do { } while (0)
#endif

__device__ __forceinline__ uint32_t pack_bf16x2(float a, float b) {
    const f32x2 v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));  // v_cvt_pk_bf16_f32 (RNE)
}
__device__ __forceinline__ float bf16_lo(uint32_t u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t u) { return __builtin_bit_cast(float, u & 0xffff0000u); }

// Split 4 consecutive channels of one row into the format's planes and store them.
template <int F>
__device__ __forceinline__ void split_store(uint16_t *planes, int plane_elems, int row, int col, float4 v) {
    uint16_t *dst = planes + row * kLDB + col;
#pragma unroll
    for (int p = 0; p < Fmt<F>::NP; ++p) {
        const uint32_t a = Fmt<F>::pack(v.x, v.y), b = Fmt<F>::pack(v.z, v.w);
        *reinterpret_cast<uint2 *>(dst + p * plane_elems) = make_uint2(a, b);
        if (p + 1 < Fmt<F>::NP) {  // the residual is exact in fp32
            const f32x2 ra = Fmt<F>::unpack(a), rb = Fmt<F>::unpack(b);
            v.x -= ra.x; v.y -= ra.y; v.z -= rb.x; v.w -= rb.y;
        }
    }
}

template <int RB, int F>
struct Tile {
    static constexpr int ROWS = 16 * RB;
    static constexpr int PLANE = ROWS * kLDB;            // bf16 elements
    static constexpr int ABUF_BYTES = Fmt<F>::NP * PLANE * 2;    // one set of planes
    static constexpr int T_BYTES = ROWS * kLDA * 4;      // fp32 epilogue tile
    static constexpr int SMEM = (2 * ABUF_BYTES > T_BYTES + ABUF_BYTES ? 2 * ABUF_BYTES : T_BYTES + ABUF_BYTES);
};

// acc[rb][cb] (16 x 16 blocks: rows 16rb.., channels 32w + 16cb..) += A(planes) * W
// A operand of 16x16x32: lane l holds A[l & 15][k = 8 (l >> 4) + j]; B operand B[k][col = l & 15].
template <int F>
struct BFrag { uint4 v[Fmt<F>::NP][2]; };

template <int F>
__device__ __forceinline__ void load_b(BFrag<F> &b, const uint4 *__restrict__ Bw, int wave, int lane, int s) {
#pragma unroll
    for (int p = 0; p < Fmt<F>::NP; ++p)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
            b.v[p][cb] = Bw[((((p * 4 + wave) * 4 + s) * 2 + cb) << 6) + lane];
}

template <int RB, int F>
__device__ __forceinline__ void kstep(const uint16_t *__restrict__ arow, int s, const BFrag<F> &b, f32x4 (&acc)[RB][2]) {
    constexpr int PLANE = Tile<RB, F>::PLANE;
    uint4 a[Fmt<F>::NP][RB];
#pragma unroll
    for (int p = 0; p < Fmt<F>::NP; ++p)
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
            a[p][rb] = *reinterpret_cast<const uint4 *>(arow + p * PLANE + rb * 16 * kLDB + 32 * s);
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            f32x4 c = acc[rb][cb];
#pragma unroll
            for (int q = 0; q < Fmt<F>::NPROD; ++q)   // smallest terms first
                c = Fmt<F>::mfma(a[Fmt<F>::PA[q]][rb], b.v[Fmt<F>::PB[q]][cb], c);
            acc[rb][cb] = c;
        }
}

// Weight-fragment ring: 4 sets = the 4 K-steps of a pass, prefetch distance 3 K-steps.  On entry
// ring.b[0..2] hold K-steps 0..2 of Bw; step s first issues the load of the step 3 ahead (K-step 3 of
// this pass for s = 0, K-step s-1 of Bw_next for s >= 1) and then runs its MFMAs, so a fragment has
// three K-steps of MFMA time to arrive and the stream stays ahead across the per-relation barrier.
template <int F>
struct BRing { BFrag<F> b[4]; };

template <int F>
__device__ __forceinline__ void ring_prime(BRing<F> &r, const uint4 *__restrict__ Bw, int wave, int lane) {
    load_b<F>(r.b[0], Bw, wave, lane, 0);
    load_b<F>(r.b[1], Bw, wave, lane, 1);
    load_b<F>(r.b[2], Bw, wave, lane, 2);
}

// Shallow variant: two fragment sets, prefetch distance 1 K-step (24 fewer VGPRs at NP = 3, which is
// what lets two 8-wave workgroups share a CU; the latency is then hidden across workgroups instead).
template <int F>
struct BPair { BFrag<F> b[2]; };

template <int F>
__device__ __forceinline__ void ring_prime(BPair<F> &r, const uint4 *__restrict__ Bw, int wave, int lane) {
    load_b<F>(r.b[0], Bw, wave, lane, 0);
}

// Minimal variant: ONE fragment set (16 fewer VGPRs at NP = 2), the next K-step's fragments are requested as soon as
// the current K-step's MFMAs are issued.  For 48-row tiles, where it is what lets two workgroups share a CU.
template <int F>
struct BOne { BFrag<F> b[1]; };

template <int F>
__device__ __forceinline__ void ring_prime(BOne<F> &r, const uint4 *__restrict__ Bw, int wave, int lane) {
    load_b<F>(r.b[0], Bw, wave, lane, 0);
}

template <int RB, int F>
__device__ __forceinline__ void gemm_pass(const uint16_t *__restrict__ A, const uint4 *__restrict__ Bw,
                                          const uint4 *__restrict__ Bw_next, BOne<F> &r, int wave, int lane,
                                          f32x4 (&acc)[RB][2]) {
    const uint16_t *arow = A + (lane & 15) * kLDB + 8 * (lane >> 4);
    kstep<RB, F>(arow, 0, r.b[0], acc);
    load_b<F>(r.b[0], Bw, wave, lane, 1);
    kstep<RB, F>(arow, 1, r.b[0], acc);
    load_b<F>(r.b[0], Bw, wave, lane, 2);
    kstep<RB, F>(arow, 2, r.b[0], acc);
    load_b<F>(r.b[0], Bw, wave, lane, 3);
    kstep<RB, F>(arow, 3, r.b[0], acc);
    if (Bw_next != nullptr) load_b<F>(r.b[0], Bw_next, wave, lane, 0);
}

template <int RB, int F>
__device__ __forceinline__ void gemm_pass(const uint16_t *__restrict__ A, const uint4 *__restrict__ Bw,
                                          const uint4 *__restrict__ Bw_next, BPair<F> &r, int wave, int lane,
                                          f32x4 (&acc)[RB][2]) {
    const uint16_t *arow = A + (lane & 15) * kLDB + 8 * (lane >> 4);
    load_b<F>(r.b[1], Bw, wave, lane, 1);
    kstep<RB, F>(arow, 0, r.b[0], acc);
    load_b<F>(r.b[0], Bw, wave, lane, 2);
    kstep<RB, F>(arow, 1, r.b[1], acc);
    load_b<F>(r.b[1], Bw, wave, lane, 3);
    kstep<RB, F>(arow, 2, r.b[0], acc);
    if (Bw_next != nullptr) load_b<F>(r.b[0], Bw_next, wave, lane, 0);
    kstep<RB, F>(arow, 3, r.b[1], acc);
}

template <int RB, int F>
__device__ __forceinline__ void gemm_pass(const uint16_t *__restrict__ A, const uint4 *__restrict__ Bw,
                                          const uint4 *__restrict__ Bw_next, BRing<F> &r, int wave, int lane,
                                          f32x4 (&acc)[RB][2]) {
    const uint16_t *arow = A + (lane & 15) * kLDB + 8 * (lane >> 4);
    load_b<F>(r.b[3], Bw, wave, lane, 3);
    kstep<RB, F>(arow, 0, r.b[0], acc);
    if (Bw_next != nullptr) load_b<F>(r.b[0], Bw_next, wave, lane, 0);
    kstep<RB, F>(arow, 1, r.b[1], acc);
    if (Bw_next != nullptr) load_b<F>(r.b[1], Bw_next, wave, lane, 1);
    kstep<RB, F>(arow, 2, r.b[2], acc);
    if (Bw_next != nullptr) load_b<F>(r.b[2], Bw_next, wave, lane, 2);
    kstep<RB, F>(arow, 3, r.b[3], acc);
}

// Second GEMM of a block (one pass over wp2).  On entry the ring holds K-step 0 (K-steps 0..2 for the deep
// ring), prefetched by the last relation pass; gemm2_prefetch, called while the row phase runs on the other waves, adds the K-steps the
// ring has room for, so that gemm2_pass starts with them landed.
template <int F>
__device__ __forceinline__ void gemm2_prefetch(BPair<F> &r, const uint4 *__restrict__ Bw, int wave, int lane) {
    load_b<F>(r.b[1], Bw, wave, lane, 1);
}
template <int F>
__device__ __forceinline__ void gemm2_prefetch(BOne<F> &, const uint4 *__restrict__, int, int) {}
template <int RB, int F>
__device__ __forceinline__ void gemm2_pass(const uint16_t *__restrict__ A, const uint4 *__restrict__ Bw, BOne<F> &r,
                                           int wave, int lane, f32x4 (&acc)[RB][2]) {
    gemm_pass<RB, F>(A, Bw, nullptr, r, wave, lane, acc);
}
template <int F>
__device__ __forceinline__ void gemm2_prefetch(BRing<F> &r, const uint4 *__restrict__ Bw, int wave, int lane) {
    load_b<F>(r.b[3], Bw, wave, lane, 3);   // K-steps 0..2 came with the last relation pass
}
template <int RB, int F>
__device__ __forceinline__ void gemm2_pass(const uint16_t *__restrict__ A, const uint4 *__restrict__ Bw, BPair<F> &r,
                                           int wave, int lane, f32x4 (&acc)[RB][2]) {
    const uint16_t *arow = A + (lane & 15) * kLDB + 8 * (lane >> 4);
    kstep<RB, F>(arow, 0, r.b[0], acc);
    load_b<F>(r.b[0], Bw, wave, lane, 2);
    kstep<RB, F>(arow, 1, r.b[1], acc);
    load_b<F>(r.b[1], Bw, wave, lane, 3);
    kstep<RB, F>(arow, 2, r.b[0], acc);
    kstep<RB, F>(arow, 3, r.b[1], acc);
}
template <int RB, int F>
__device__ __forceinline__ void gemm2_pass(const uint16_t *__restrict__ A, const uint4 *__restrict__, BRing<F> &r,
                                           int wave, int lane, f32x4 (&acc)[RB][2]) {
    const uint16_t *arow = A + (lane & 15) * kLDB + 8 * (lane >> 4);
    kstep<RB, F>(arow, 0, r.b[0], acc);
    kstep<RB, F>(arow, 1, r.b[1], acc);
    kstep<RB, F>(arow, 2, r.b[2], acc);
    kstep<RB, F>(arow, 3, r.b[3], acc);
}

// C/D layout of 16x16: col = lane & 15, row = 4 (lane >> 4) + reg
template <int RB>
__device__ __forceinline__ void acc_store(float *T, const f32x4 (&acc)[RB][2], int lane, int wave) {
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            float *p = T + (16 * rb + 4 * (lane >> 4)) * kLDA + 32 * wave + 16 * cb + (lane & 15);
#pragma unroll
            for (int i = 0; i < 4; ++i) p[i * kLDA] = acc[rb][cb][i];
        }
}

template <int RB>
__device__ __forceinline__ void acc_zero(f32x4 (&acc)[RB][2]) {
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) acc[rb][cb] = f32x4{0.f, 0.f, 0.f, 0.f};
}

template <int F>
__device__ __forceinline__ void row_split_store(uint16_t *planes, int plane_elems, int row, int t, const RowVals &r) {
#pragma unroll
    for (int j = 0; j < 4; ++j) split_store<F>(planes, plane_elems, row, 4 * (t & 7) + 32 * j, r.v[j]);
}

// ------------------------------------------------------- shared pieces -----
// h1[row][c] = ReLU(w1[c][0] x + w1[c][1] y + b1[c]) for the thread's 16 channels, split into planes
template <int F>
__device__ __forceinline__ void lin2_relu_split(uint16_t *planes, int plane_elems, int row, int t, float x, float y,
                                                const float *__restrict__ w1, const float *__restrict__ b1) {
    const int c0 = 4 * (t & 7);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = c0 + 32 * j;
        const float4 wa = *reinterpret_cast<const float4 *>(w1 + 2 * c);
        const float4 wb = *reinterpret_cast<const float4 *>(w1 + 2 * c + 4);
        const float4 bb = *reinterpret_cast<const float4 *>(b1 + c);
        float4 o;
        o.x = relu_nan(x * wa.x + y * wa.y + bb.x);
        o.y = relu_nan(x * wa.z + y * wa.w + bb.y);
        o.z = relu_nan(x * wb.x + y * wb.y + bb.z);
        o.w = relu_nan(x * wb.z + y * wb.w + bb.w);
        split_store<F>(planes, plane_elems, row, c, o);
    }
}


}  // namespace lgcn
