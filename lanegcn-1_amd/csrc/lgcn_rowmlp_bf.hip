// Split-precision variants of the row-block kernels (LGCN_MMA_BF16X3 / LGCN_MMA_F16X2 / LGCN_MMA_BF16).
//
// fp32 operands are split into 16-bit planes (struct Fmt below) and a K = 128 pass contracts the
// planes pairwise on v_mfma_f32_16x16x32_{bf16,f16} with fp32 accumulation:
//   bf16x3: x = hi + mid + lo (3 x 8 bits), 6 products    f16x2: x = hi + lo (2 x 11 bits), 3 products
// Tile = 16*RB rows (RB = 1..4 CSR sub-tiles, picked per launch so that the tile
// count fits the 256 CUs in as few rounds as possible) x 128 output channels;
// wave w of the 4 MFMA waves owns channels [32w, 32w+32) as two 16-column
// blocks and streams its own packed weight slice L2 -> VGPR; the A planes live
// in LDS (272-byte rows: conflict-free ds_read_b128), double-buffered against
// the 4 gather waves exactly like the f32 kernel.
#include "lgcn_common.hpp"
#include "lgcn_tile.hpp"
#include "lgcn_mma_bf.hpp"
#include <cstdlib>
#include <type_traits>

#ifndef LGCN_RB_ONE
#define LGCN_RB_ONE 3      // tile height (in 16-row blocks) built with the one-set weight ring and held to 128 VGPRs
#endif

namespace lgcn {


// ------------------------------------------------------------ packing -----
template <int F>
__device__ __forceinline__ void pack_block_bf(const float *__restrict__ W, int ld, uint16_t *__restrict__ out, int transpose,
                                              int i) {
    // one thread per (w, s, cb, lane): 8 consecutive k of one weight row -> 8 elements per plane
    if (i >= 4 * 4 * 2 * 64) return;
    const int lane = i & 63, cb = (i >> 6) & 1, s = (i >> 7) & 3, w = i >> 9;
    const int orow = 32 * w + 16 * cb + (lane & 15), k0 = 32 * s + 8 * (lane >> 4);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = transpose ? W[(int64_t)(k0 + j) * ld + orow] : W[(int64_t)orow * ld + k0 + j];
#pragma unroll
    for (int p = 0; p < Fmt<F>::NP; ++p) {
        uint32_t q[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            q[j] = Fmt<F>::pack(v[2 * j], v[2 * j + 1]);
            const f32x2 r = Fmt<F>::unpack(q[j]);
            v[2 * j] -= r.x;
            v[2 * j + 1] -= r.y;
        }
        uint4 *dst = reinterpret_cast<uint4 *>(out) + ((((p * 4 + w) * 4 + s) * 2 + cb) << 6) + lane;
        *dst = make_uint4(q[0], q[1], q[2], q[3]);
    }
}

template <int F>
__global__ __launch_bounds__(256) void k_pack_weight_bf(const float *__restrict__ W, int ld, uint16_t *__restrict__ out,
                                                        int transpose) {
    pack_block_bf<F>(W, ld, out, transpose, blockIdx.x * blockDim.x + threadIdx.x);
}

template <int F>
__global__ __launch_bounds__(256) void k_pack_weight_batch_bf(const lgcn_pack_job_t *__restrict__ jobs) {
    const lgcn_pack_job_t job = jobs[blockIdx.y];
    pack_block_bf<F>(job.W, job.ld, reinterpret_cast<uint16_t *>(job.out), job.transpose, blockIdx.x * blockDim.x + threadIdx.x);
}

// ------------------------------------------------------------ agg_mlp -----
// Per-tile index slice, loaded into LDS once: the rowptr chunk of every CSR sub-tile (n_rel_csr*16+1
// ints each, contiguous in the tile-major plan), the col entries they span, and rowptr[row0..row0+ROWS]
// for the RANGE relation.  A gather then needs ONE dependent global access (the source row) per pass
// instead of three (rowptr -> col -> row).
template <int RB>
struct TileIdx {
    static constexpr int RP = LGCN_MAX_REL * 16 + 4;   // ints per sub-tile chunk (<= 16*16+1)
    static constexpr int COLCAP = 64 * 16 * RB;        // col entries kept in LDS (fallback: global)
    static constexpr int INTS = RB * RP + COLCAP + 16 * RB + 4;
    int *rp;      // [RB][RP]
    int *col;     // [COLCAP]
    int *rng;     // [16*RB + 1]
    __device__ explicit TileIdx(int *base) { rp = base; col = rp + RB * RP; rng = col + COLCAP; }
};

// One relation's A operand: row n of the tile = sum of the source rows of n under this relation, split into
// the format's planes.  One half-wave per destination row (32 lanes x 16 B = one 512-B source row), 8 rows per
// sweep over the 4 gather waves, 2 RB sweeps.  All first-edge loads of the sweeps are issued before any is
// used, and the second-edge loads of rows that have one go out with them, so a pass costs ONE global round
// trip for in-degree <= 2 (the common case: lane graphs branch rarely) and one more per further edge of the
// widest row, instead of one per sweep.  Edges are summed in index order (as the f32 kernel and a sequential
// index_add_ do).
// LDSCOL: the tile's col entries are in LDS (ix.col, local index = global index + cadj[sub-tile]).
// MODE is a template parameter so that each mode's index reads and row loads are issued as straight-line batches.
//
// Registers: a wave is either a gather wave or an MFMA wave for the whole kernel, but the allocator sees one
// function, so what the gather holds would come on top of the accumulators and weight fragments that only
// the MFMA waves use.  The gather therefore keeps its 2 RB row sums IN the accumulator variables (same
// count: acc[RB][2]) and, where they fit, the second-edge rows in the spare weight-fragment set; both are
// dead values on a gather wave.  This is what keeps 32-row tiles at 128 VGPRs (two workgroups per CU).
template <int F, class Ring, int IT>
struct XRows {   // second-edge rows: the first slots live in the ring's LAST fragment set, the rest in a local array
    static constexpr int kSets = sizeof(Ring) / sizeof(BFrag<F>);
    static constexpr int kSlots = 2 * Fmt<F>::NP;                 // 16-B slots of one fragment set
    static constexpr int kOwn = IT > kSlots ? IT - kSlots : 0;
    Ring &ring;
    f32x4 own[kOwn > 0 ? kOwn : 1];
    __device__ explicit XRows(Ring &r) : ring(r) {}
    __device__ __forceinline__ void set(int i, f32x4 v) {
        if (i < kSlots) ring.b[kSets - 1].v[i >> 1][i & 1] = __builtin_bit_cast(uint4, v);
        else own[i - kSlots] = v;
    }
    __device__ __forceinline__ f32x4 get(int i) const {
        if (i < kSlots) return __builtin_bit_cast(f32x4, ring.b[kSets - 1].v[i >> 1][i & 1]);
        return own[i - kSlots];
    }
};

template <int RB, int F, int MODE, bool LDSCOL, class Ring>
__device__ __forceinline__ void gather_mode(uint16_t *__restrict__ Abuf, const lgcn_agg_mlp_t &p, int ri, int tile, int gt,
                                            const TileIdx<RB> &ix, const int (&cadj)[RB], f32x4 (&s)[RB][2], Ring &ring) {
    constexpr int IT = 2 * RB;
    constexpr int mode = MODE;
    const f32x4 *__restrict__ src = reinterpret_cast<const f32x4 *>(p.rel[ri].src);
    const int ridx = p.rel[ri].ridx;     // wave-uniform
    const int hw = gt >> 5, l = gt & 31;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    int b[IT], e[IT];
    if (mode == LGCN_REL_CSR) {
        const int *rp = ix.rp + ridx * 16 + hw;
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int o = (it >> 1) * TileIdx<RB>::RP + (it & 1) * 8;
            const int adj = LDSCOL ? cadj[it >> 1] : 0;
            b[it] = rp[o] + adj;
            e[it] = rp[o + 1] + adj;
        }
    } else if (mode == LGCN_REL_RANGE || mode == LGCN_REL_RANGE16) {
#pragma unroll
        for (int it = 0; it < IT; ++it) { b[it] = ix.rng[it * 8 + hw]; e[it] = ix.rng[it * 8 + hw + 1]; }
    } else {
        // an IDENT gather is the same every pass; the opaque zero keeps its addressing out of the loop
        // pre-header, where the hoisted values would hold registers for the whole kernel
        int opq = 0;
        asm volatile("" : "+v"(opq));
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int64_t n = (int64_t)tile * (16 * RB) + it * 8 + hw;
            const bool live = n < p.n_rows;
            b[it] = live ? (int)n + opq : 0;
            e[it] = live ? (int)n + opq + 1 : 0;
        }
    }
    // source row of edge j (valid j only)
    auto row_of = [&](int j) -> unsigned { return mode != LGCN_REL_CSR ? (unsigned)j : (unsigned)(LDSCOL ? ix.col[j] : p.col[j]); };
    auto load = [&](unsigned r) -> f32x4 { return src[((uint64_t)r << 5) + l]; };
    // edges j .. e-1 of one row added to acc in index order, four loads in flight
    // RANGE16: the rows of a segment [b, e) that hold data are b and the multiples of 16 inside (b, e)
    // (lgcn_att_pairs_ws, seg = 16: per-target sums of 16-aligned pieces, written at each piece's first row)
    constexpr bool STEP16 = MODE == LGCN_REL_RANGE16;
    auto nxt = [&](int j) -> int { return STEP16 ? ((j >> 4) + 1) << 4 : j + 1; };
    auto tail = [&](f32x4 acc, int j, int end) -> f32x4 {
        if (STEP16) {
            for (; j < end; j += 16) acc = acc + load((unsigned)j);
            return acc;
        }
        if (MODE == LGCN_REL_RANGE && RB == 1) {    // Att segment sums (tens of rows per target): eight in flight where registers allow
            for (; j + 7 < end; j += 8) {
                f32x4 y[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) y[q] = load(row_of(j + q));
#pragma unroll
                for (int q = 0; q < 8; ++q) acc = acc + y[q];
            }
        }
        for (; j + 3 < end; j += 4) {
            const f32x4 y0 = load(row_of(j)), y1 = load(row_of(j + 1)), y2 = load(row_of(j + 2)), y3 = load(row_of(j + 3));
            acc = (((acc + y0) + y1) + y2) + y3;
        }
        for (; j < end; ++j) acc = acc + load(row_of(j));
        return acc;
    };

    XRows<F, Ring, IT> x(ring);
    bool more = false;
#pragma unroll
    for (int it = 0; it < IT; ++it) more = more | (nxt(b[it]) < e[it]);
    const bool any2 = MODE != LGCN_REL_IDENT && __any(more);   // some row of this wave has a second edge
    unsigned r0[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        r0[it] = 0u;
        if (b[it] < e[it]) r0[it] = row_of(b[it]);
    }
#pragma unroll
    for (int it = 0; it < IT; ++it) s[it >> 1][it & 1] = load(r0[it]);   // unconditional (row 0 when the row has no source)
    if (any2) {
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            f32x4 v = zero;
            if (nxt(b[it]) < e[it]) v = load(row_of(nxt(b[it])));
            x.set(it, v);
        }
    }
#pragma unroll
    for (int it = 0; it < IT; ++it) s[it >> 1][it & 1] = b[it] < e[it] ? s[it >> 1][it & 1] : zero;
    if (any2) {
#pragma unroll
        for (int it = 0; it < IT; ++it) s[it >> 1][it & 1] = s[it >> 1][it & 1] + x.get(it);
        more = false;
#pragma unroll
        for (int it = 0; it < IT; ++it) more = more | (nxt(nxt(b[it])) < e[it]);
        if (__any(more)) {   // in-degree > 2 (rare in lane graphs, the rule for Att's RANGE sums)
#pragma unroll
            for (int it = 0; it < IT; ++it) s[it >> 1][it & 1] = tail(s[it >> 1][it & 1], nxt(nxt(b[it])), e[it]);
        }
    }
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const f32x4 v = s[it >> 1][it & 1];
        split_store<F>(Abuf, Tile<RB, F>::PLANE, it * 8 + hw, 4 * l, make_float4(v[0], v[1], v[2], v[3]));
    }
}

// KIND 1 launches (LaneConv) hold IDENT and CSR relations, KIND 0 launches IDENT and RANGE ones.
template <int RB, int F, int KIND, bool LDSCOL, class Ring>
__device__ __forceinline__ void gather_rel(uint16_t *__restrict__ Abuf, const lgcn_agg_mlp_t &p, int ri, int tile, int gt,
                                           const TileIdx<RB> &ix, const int (&cadj)[RB], f32x4 (&s)[RB][2], Ring &ring) {
    const int mode = p.rel[ri].mode;     // wave-uniform
    if (mode == LGCN_REL_IDENT) gather_mode<RB, F, LGCN_REL_IDENT, false>(Abuf, p, ri, tile, gt, ix, cadj, s, ring);
    else if (KIND == 1) gather_mode<RB, F, LGCN_REL_CSR, LDSCOL>(Abuf, p, ri, tile, gt, ix, cadj, s, ring);
    else if (mode == LGCN_REL_RANGE16) gather_mode<RB, F, LGCN_REL_RANGE16, false>(Abuf, p, ri, tile, gt, ix, cadj, s, ring);
    else gather_mode<RB, F, LGCN_REL_RANGE, false>(Abuf, p, ri, tile, gt, ix, cadj, s, ring);
}

template <int RB, int F, int KIND, bool DEEP>
__device__ __forceinline__ void agg_body(const lgcn_agg_mlp_t &p, int n_tiles, int bid, unsigned char *smem) {
    using TL = Tile<RB, F>;
    using IX = TileIdx<RB>;
    constexpr int ROWS = TL::ROWS;
    uint16_t *buf0 = reinterpret_cast<uint16_t *>(smem);
    uint16_t *buf1 = reinterpret_cast<uint16_t *>(smem + TL::ABUF_BYTES);
    float *T = reinterpret_cast<float *>(smem);                                 // aliases the A buffers
    uint16_t *Yp = reinterpret_cast<uint16_t *>(smem + TL::T_BYTES);            // stage-2 operand planes
    int *act = reinterpret_cast<int *>(smem + TL::SMEM);                        // [0..15] ids, [16] count
    const IX ix(reinterpret_cast<int *>(smem + TL::SMEM + 128));

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile = xcd_chunk_remap(bid, n_tiles);
    const int64_t row0 = (int64_t)tile * ROWS;
    const int64_t n_sub = (p.n_rows + 15) >> 4;
#ifdef LGCN_STAMPS
    // wave 0 (MFMA role) -> slots 0..63, wave 4 (gather role) -> slots 64..127 of this block's 128-slot record
    unsigned long long *sbuf = nullptr;
    if (KIND == 1 && p.out_pre && (wave == 0 || wave == 4))
        sbuf = reinterpret_cast<unsigned long long *>(p.out_pre) + (int64_t)bid * 128 + (wave == 4 ? 64 : 0);
    LGCN_STAMP(0);
#endif
    const int nrc = p.n_rel_csr;
    const bool has_csr = KIND == 1, has_rng = KIND == 0 && p.rowptr != nullptr;

    const int flags = p.flags;
    const bool two = (flags & LGCN_F_GEMM2) != 0;
    const int gt = tid - 256;
    typename std::conditional<DEEP, BRing<F>, typename std::conditional<RB == LGCN_RB_ONE && F != 0, BOne<F>, BPair<F>>::type>::type bfrag;
    f32x4 acc[RB][2];       // MFMA waves: accumulators; gather waves: the row sums of the relation in flight
    int cadj[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) cadj[rb] = 0;

    // ---- prologue, one barrier.  Order of issue: (1) the index words every wave needs from global memory
    // (bounds of the tile's col entries, this thread's word of each rowptr chunk, wave 7: the activity words
    // of its relation), (2) relation 0 when it is the row itself (LaneConv's ctr, every Linear: always
    // active, needs no index): its rows on the gather waves, its first weight fragments on the MFMA waves,
    // (3) the col entries, whose addresses depend on (1).  (1) and (2) share one round trip.
    float *gnp = reinterpret_cast<float *>(smem + TL::SMEM + 128 + IX::INTS * 4);   // gn1 g|b, gn2 g|b, chained query g|b
    const bool chain_u = KIND == 0 && p.ch_wu != nullptr, chain_v = KIND == 0 && p.ch_wv != nullptr;
    if (tid >= 128 && tid < 256) {
        const int c = tid - 128;
        if (flags & LGCN_F_GN1) { gnp[c] = p.gn1_g[c]; gnp[kC + c] = p.gn1_b[c]; }
        if (flags & LGCN_F_GN2) { gnp[2 * kC + c] = p.gn2_g[c]; gnp[3 * kC + c] = p.gn2_b[c]; }
        if (chain_u) { gnp[4 * kC + c] = p.ch_gq_g[c]; gnp[5 * kC + c] = p.ch_gq_b[c]; }
    }
    const int len = nrc * 16 + 1;     // <= 257 ints per sub-tile: one per thread
    int c0[RB], c1[RB], rpv[RB], a0[RB], a1[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) c0[rb] = c1[rb] = rpv[rb] = a0[rb] = a1[rb] = 0;
    if (has_csr) {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            const int64_t sub = (int64_t)tile * RB + rb;
            const int32_t *chunk = p.rowptr + sub * nrc * 16;
            if (sub < n_sub) {
                c0[rb] = chunk[0];
                c1[rb] = chunk[nrc * 16];
                if (tid < len) rpv[rb] = chunk[tid];
                if (wave == 7 && lane < p.n_rel && p.rel[lane].mode == LGCN_REL_CSR) {
                    a0[rb] = chunk[p.rel[lane].ridx * 16];
                    a1[rb] = chunk[p.rel[lane].ridx * 16 + 16];
                }
            }
        }
    } else if (has_rng) {
        if (wave == 7 && lane < p.n_rel && (p.rel[lane].mode == LGCN_REL_RANGE || p.rel[lane].mode == LGCN_REL_RANGE16)) {
            a0[0] = p.rowptr[row0 < p.n_rows ? row0 : p.n_rows];
            a1[0] = p.rowptr[row0 + ROWS < p.n_rows ? row0 + ROWS : p.n_rows];
        }
    }
    const bool early = p.rel[0].mode == LGCN_REL_IDENT;
    if (early) {
        if (wave >= 4) gather_mode<RB, F, LGCN_REL_IDENT, false>(buf0, p, 0, tile, gt, ix, cadj, acc, bfrag);
        else ring_prime<F>(bfrag, reinterpret_cast<const uint4 *>(p.rel[0].wp), wave, lane);
    }
    bool lds_col = false;
    if (has_csr) {
        int lo = 0;
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            c0[rb] = __builtin_amdgcn_readfirstlane(c0[rb]);
            c1[rb] = __builtin_amdgcn_readfirstlane(c1[rb]);
            cadj[rb] = lo - c0[rb];               // global col index -> index into ix.col
            lo += c1[rb] - c0[rb];
        }
        lds_col = lo <= IX::COLCAP;
        if (lds_col) {
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
                for (int j = tid; j < c1[rb] - c0[rb]; j += 512) ix.col[c0[rb] + cadj[rb] + j] = p.col[c0[rb] + j];
        }
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
            if (tid < len) ix.rp[rb * IX::RP + tid] = rpv[rb];
    } else if (has_rng) {
        for (int j = tid; j <= ROWS; j += 512) {
            const int64_t n = row0 + j < p.n_rows ? row0 + j : p.n_rows;
            ix.rng[j] = p.rowptr[n];
        }
    }
    if (wave == 7) {     // active relations of this tile, in relation order
        bool on = false;
        if (lane < p.n_rel) {
            on = p.rel[lane].mode == LGCN_REL_IDENT;
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) on = on | (a1[rb] > a0[rb]);
        }
        const unsigned long long m = __ballot(on);
        if (on) act[__popcll(m & ((1ull << lane) - 1ull))] = lane;
        if (lane == 0) act[16] = __popcll(m);
    }
    lds_barrier();
    const int nact = __builtin_amdgcn_readfirstlane(act[16]);
    LGCN_STAMP(1);

    // ---- main loop: MFMA waves run pass i on buffer i & 1 while the gather waves fill the other buffer
    // with relation i + 1 (one barrier per relation).  Latency is hidden across workgroups (keep the
    // VGPR count <= 128 so that two 8-wave workgroups fit a CU): a deeper per-wave register ring was
    // measured slower because it halves that occupancy.
    auto rel_at = [&](int i) { return __builtin_amdgcn_readfirstlane(act[i]); };
    auto gather = [&](uint16_t *dst, int ri) {
        if (KIND == 1 && lds_col) gather_rel<RB, F, KIND, true>(dst, p, ri, tile, gt, ix, cadj, acc, bfrag);
        else gather_rel<RB, F, KIND, false>(dst, p, ri, tile, gt, ix, cadj, acc, bfrag);
    };
    if (!early) {
        if (wave >= 4) {
            if (nact > 0) gather(buf0, rel_at(0));
        } else {
            const float *w0 = nact > 0 ? p.rel[rel_at(0)].wp : p.wp2;
            if (w0 != nullptr) ring_prime<F>(bfrag, reinterpret_cast<const uint4 *>(w0), wave, lane);
        }
        LGCN_STAMP(2);
        lds_barrier();
    }
    LGCN_STAMP(3);

    acc_zero<RB>(acc);
    for (int i = 0; i < nact; ++i) {
        uint16_t *cur = (i & 1) ? buf1 : buf0;
        uint16_t *nxt = (i & 1) ? buf0 : buf1;
        if (wave < 4) {
#ifdef LGCN_ABLATE   // diagnostic build only (make ablate, tools/bench_agg.py): flag bit 9 skips the MFMA passes
            if (!(flags & (1 << 9)))
#endif
            {
            const float *wn = i + 1 < nact ? p.rel[rel_at(i + 1)].wp : (two ? p.wp2 : nullptr);
            gemm_pass<RB, F>(cur, reinterpret_cast<const uint4 *>(p.rel[rel_at(i)].wp),
                              reinterpret_cast<const uint4 *>(wn), bfrag, wave, lane, acc);
            }
        } else if (i + 1 < nact
#ifdef LGCN_ABLATE   // diagnostic build only: flag bit 8 skips the in-loop gathers
                   && !(flags & (1 << 8))
#endif
        ) {
            gather(nxt, rel_at(i + 1));
        }
        LGCN_STAMP(4 + 2 * i);       // own work of pass i done
        lds_barrier();
        LGCN_STAMP(5 + 2 * i);       // barrier passed
    }

    // ---- epilogue.  The MFMA waves own the accumulators and the second GEMM; the row phases (GroupNorm,
    // residual, ReLU, stores) of the first 32 rows run on waves 4..7, which have finished gathering: the residual
    // rows are requested before the first row phase and arrive under it and the second GEMM, and the MFMA waves
    // fetch the second weight's fragments before they normalise the rows above 32 (if the tile has any).
    constexpr int NCH = (ROWS + 31) / 32;       // 32-row chunks of the row phase (8 threads per row)
    const int rt = (tid - 256) & 255;           // row-phase thread id (masked: lets the compiler drop row < ROWS)
    if (wave < 4) {
        if (p.w4 != nullptr) {  // rank-4 fp32 update: the 4 meta channels of A2M.meta
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                const float4 wc = *reinterpret_cast<const float4 *>(p.w4 + 4 * (32 * wave + 16 * cb + (lane & 15)));
#pragma unroll
                for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int64_t n = row0 + 16 * rb + 4 * (lane >> 4) + i;
                        if (n < p.n_rows) {
                            const float2 tu = reinterpret_cast<const float2 *>(p.x4_a)[n];
                            acc[rb][cb][i] += tu.x * wc.x + tu.y * wc.y + p.x4_b[n] * wc.z + p.x4_c[n] * wc.w;
                        }
                    }
            }
        }
        acc_store<RB>(T, acc, lane, wave);
    }
    LGCN_STAMP(40);
    lds_barrier();
    LGCN_STAMP(41);

    // Row-phase ownership: rows 0..31 of the tile belong to waves 4..7 (thread rt -> row rt >> 3), rows 32.. (tiles
    // of 48 / 64 rows) to the MFMA waves, which would otherwise idle through both row phases: each thread owns ONE
    // row chunk and both halves run side by side.
    const bool upper = wave < 4;                                   // this thread serves the rows above 32
    const int my_rt = upper ? tid : rt;
    const int my_row = (upper ? 32 : 0) + (my_rt >> 3);
    const bool my_has_row = (upper ? NCH > 1 : true) && my_row < ROWS;      // 16-row tiles: half of waves 4..7 idle
    const int64_t my_n = row0 + my_row;
    const bool my_live = my_has_row && my_n < p.n_rows;
    RowVals resv;     // residual row: requested here, used after GroupNorm (one stage) or after the second GEMM
    // Chained outputs (lgcn.h: ch_*): the block's final rows y go back to LDS as operand planes instead of only to HBM:
    //   u_out = ReLU(GN_q(y W_q^T)) W_u^T  (three more barriers-separated passes), v_out = y W_v^T (stored from the
    //   accumulators: 64-byte row segments, by the otherwise idle MFMA waves).  Yp is free here: its last readers (the
    //   second GEMM, or nobody) passed the barrier in front of the final row phase.
    auto chained = [&](const RowVals &y) {
        if (KIND != 0) return;
        const uint4 *wq = reinterpret_cast<const uint4 *>(chain_u ? p.ch_wq : p.ch_wv);
        if (my_has_row) row_split_store<F>(Yp, TL::PLANE, my_row, my_rt, y);
        if (wave < 4) ring_prime<F>(bfrag, wq, wave, lane);
        lds_barrier();
        if (wave < 4) {
            if (chain_u) {
                acc_zero<RB>(acc);
                gemm_pass<RB, F>(Yp, wq, reinterpret_cast<const uint4 *>(chain_v ? p.ch_wv : p.ch_wu), bfrag, wave, lane, acc);
                acc_store<RB>(T, acc, lane, wave);      // T's readers (the final row phase) wrote Yp after reading it
            }
            if (chain_v) {
                acc_zero<RB>(acc);
                gemm_pass<RB, F>(Yp, reinterpret_cast<const uint4 *>(p.ch_wv), reinterpret_cast<const uint4 *>(chain_u ? p.ch_wu : nullptr),
                                 bfrag, wave, lane, acc);
#pragma unroll
                for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int64_t n = row0 + 16 * rb + 4 * (lane >> 4) + i;
                            if (n < p.n_rows) p.ch_v_out[n * kC + 32 * wave + 16 * cb + (lane & 15)] = acc[rb][cb][i];
                        }
            }
        }
        if (!chain_u) return;
        lds_barrier();
        if (my_has_row) {
            RowVals r = row_load(T + (upper ? 32 : 0) * kLDA, my_rt);
            row_gn(r, my_rt, gnp + 4 * kC, gnp + 5 * kC, p.eps);
            row_relu(r);
            row_split_store<F>(Yp, TL::PLANE, my_row, my_rt, r);      // Yp's readers passed the barrier above
        }
        lds_barrier();
        if (wave < 4) {
            acc_zero<RB>(acc);
            gemm_pass<RB, F>(Yp, reinterpret_cast<const uint4 *>(p.ch_wu), nullptr, bfrag, wave, lane, acc);
            acc_store<RB>(T, acc, lane, wave);
        }
        lds_barrier();
        if (my_live) row_store_global(p.ch_u_out + my_n * kC, my_rt, row_load(T + (upper ? 32 : 0) * kLDA, my_rt));
    };
    if (two && upper) gemm2_prefetch<F>(bfrag, reinterpret_cast<const uint4 *>(p.wp2), wave, lane);
    if (my_has_row) {
        {   // branch-free (row clamped, p.out read and ignored without a residual): behind a branch or a
            // predicate the compiler waits for these loads at the join, i.e. before the row phase starts
            const float *rbase = (flags & LGCN_F_RES) ? p.res : p.out;
            const float *rp_ = rbase + (my_live ? my_n : 0) * kC + 4 * (my_rt & 7);
#pragma unroll
            for (int j = 0; j < 4; ++j) resv.v[j] = *reinterpret_cast<const float4 *>(rp_ + 32 * j);
        }
        RowVals r = row_load(T + (upper ? 32 : 0) * kLDA, my_rt);
#ifndef LGCN_STAMPS
        if (my_live && p.out_pre) row_store_global(p.out_pre + my_n * kC, my_rt, r);
#endif
        if (flags & LGCN_F_GN1) row_gn(r, my_rt, gnp, gnp + kC, p.eps);
        if (!two && my_live && (flags & LGCN_F_RES)) row_add(r, resv);
        if (flags & LGCN_F_RELU1) row_relu(r);
        if (two && my_live && p.out_mid) row_store_global(p.out_mid + my_n * kC, my_rt, r);
        if (two) row_split_store<F>(Yp, TL::PLANE, my_row, my_rt, r);
        else if (my_live) row_store_global(p.out + my_n * kC, my_rt, r);
        if (!two && (chain_u || chain_v)) resv = r;      // the final rows (the residual is spent)
    }
    if (!two) {
        if (chain_u || chain_v) chained(resv);
        return;
    }
    LGCN_STAMP(42);
    lds_barrier();
    LGCN_STAMP(43);
    if (wave < 4) {
        acc_zero<RB>(acc);
        gemm2_pass<RB, F>(Yp, reinterpret_cast<const uint4 *>(p.wp2), bfrag, wave, lane, acc);
        acc_store<RB>(T, acc, lane, wave);   // T and Yp are disjoint; T's readers passed the barrier above
    }
    LGCN_STAMP(44);
    lds_barrier();
    LGCN_STAMP(45);
    if (my_has_row) {
        RowVals r = row_load(T + (upper ? 32 : 0) * kLDA, my_rt);
        if (my_live && p.out_pre2) row_store_global(p.out_pre2 + my_n * kC, my_rt, r);
        if (flags & LGCN_F_GN2) row_gn(r, my_rt, gnp + 2 * kC, gnp + 3 * kC, p.eps);
        if (my_live && (flags & LGCN_F_RES)) row_add(r, resv);
        if (flags & LGCN_F_RELU2) row_relu(r);
        if (my_live) row_store_global(p.out + my_n * kC, my_rt, r);
        resv = r;
    }
    LGCN_STAMP(46);
    if (chain_u || chain_v) chained(resv);
}

// Tiles of <= 32 rows must keep two 8-wave workgroups per CU (4 waves per SIMD = 128 VGPRs): that co-residency
// is what hides the weight-fragment latency; the register allocator is held to it (also where that costs a few
// spilled registers: three-plane bf16x3 at RB = 2 runs 52 us held to 128 VGPRs, 74 us left free).
// (Round 3: the three-plane kernels are no longer held to 128 VGPRs -- they spilled 10-170 registers to scratch there;
// the library ships no kernel that uses scratch, tests/test_host_cabi.py.  bf16x3 row blocks run one workgroup per CU.)
#define LGCN_WAVES_PER_SIMD(RB_, DEEP_, F_) \
    __attribute__((amdgpu_waves_per_eu((F_) != 0 && ((RB_) <= 2 || (RB_) == LGCN_RB_ONE) && !(DEEP_) ? 4 : 2)))

template <int RB, int F, int KIND, bool DEEP>
__global__ __launch_bounds__(512) LGCN_WAVES_PER_SIMD(RB, DEEP, F) void k_agg_mlp_bf(const lgcn_agg_mlp_t p, int n_tiles) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[Tile<RB, F>::SMEM + 128 + TileIdx<RB>::INTS * 4 + 6 * kC * 4];
    agg_body<RB, F, KIND, DEEP>(p, n_tiles, blockIdx.x, smem);
}

// Two independent row blocks in one launch (Att's U and V: same shape of work, different inputs): blocks
// [0, tiles_a) run problem a, the rest problem b.  One kernel boundary and one launch latency instead of two.
template <int RB, int F>
__global__ __launch_bounds__(512) LGCN_WAVES_PER_SIMD(RB, false, F) void k_agg_mlp_bf2(const lgcn_agg_mlp_t pa, const lgcn_agg_mlp_t pb, int tiles_a,
                                                     int tiles_b) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[Tile<RB, F>::SMEM + 128 + TileIdx<RB>::INTS * 4 + 6 * kC * 4];
    if ((int)blockIdx.x < tiles_a) agg_body<RB, F, 0, false>(pa, tiles_a, blockIdx.x, smem);
    else agg_body<RB, F, 0, false>(pb, tiles_b, blockIdx.x - tiles_a, smem);
}

// Up to LGCN_MAX_MULTI independent row blocks in one launch (lgcn_agg_mlp_multi): the problems' tiles follow each other
// in the grid; a workgroup reads its problem's arguments from the kernel-argument segment by index (scalar loads).
struct MultiArgs {
    lgcn_agg_mlp_t p[LGCN_MAX_MULTI];
    int tiles[LGCN_MAX_MULTI];
    int n;
};
template <int RB, int F>
__global__ __launch_bounds__(512) LGCN_WAVES_PER_SIMD(RB, false, F) void k_agg_mlp_bfn(const MultiArgs m) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[Tile<RB, F>::SMEM + 128 + TileIdx<RB>::INTS * 4 + 6 * kC * 4];
    int b = blockIdx.x, i = 0;
    while (i + 1 < m.n && b >= m.tiles[i]) { b -= m.tiles[i]; ++i; }
    agg_body<RB, F, 0, false>(m.p[i], m.tiles[i], b, smem);
}

template <int RB, int F>
__global__ __launch_bounds__(256) void k_mapnet_input_bf(const InputParams p, int n_tiles) {
    using TL = Tile<RB, F>;
    constexpr int ROWS = TL::ROWS;
    __shared__ __attribute__((aligned(16))) unsigned char smem[TL::ABUF_BYTES + TL::T_BYTES];
    uint16_t *A = reinterpret_cast<uint16_t *>(smem);
    float *T = reinterpret_cast<float *>(smem + TL::ABUF_BYTES);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t row0 = (int64_t)blockIdx.x * ROWS;
    f32x4 acc[RB][2];
    RowVals keep[(ROWS + 31) / 32];

#pragma unroll
    for (int br = 0; br < 2; ++br) {
        const float2 *xy = reinterpret_cast<const float2 *>(br == 0 ? p.ctrs : p.feats);
        const float *w1 = br == 0 ? p.wa1 : p.ws1, *b1 = br == 0 ? p.ba1 : p.bs1;
        const float *wp = br == 0 ? p.wpa2 : p.wps2;
        const float *g = br == 0 ? p.ga : p.gs, *bt = br == 0 ? p.bta : p.bts;
#pragma unroll
        for (int c0 = 0; c0 < ROWS; c0 += 32) {
            const int row = c0 + (tid >> 3);
            if (row < ROWS) {
                float2 v = make_float2(0.f, 0.f);
                if (row0 + row < p.n_rows) v = xy[row0 + row];
                lin2_relu_split<F>(A, TL::PLANE, row, tid, v.x, v.y, w1, b1);
            }
        }
        lds_barrier();
        acc_zero<RB>(acc);
        {
            BPair<F> bf;
            ring_prime<F>(bf, reinterpret_cast<const uint4 *>(wp), wave, lane);
            gemm_pass<RB, F>(A, reinterpret_cast<const uint4 *>(wp), nullptr, bf, wave, lane, acc);
        }
        acc_store<RB>(T, acc, lane, wave);
        lds_barrier();
#pragma unroll
        for (int c0 = 0; c0 < ROWS; c0 += 32) {
            const int row = c0 + (tid >> 3);
            if (row < ROWS) {
                RowVals r = row_load(T + c0 * kLDA, tid);
                row_gn(r, tid, g, bt, p.eps);
                if (br == 0) {
                    keep[c0 / 32] = r;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) r.v[j] = f4add(r.v[j], keep[c0 / 32].v[j]);
                    row_relu(r);
                    if (row0 + row < p.n_rows) row_store_global(p.out + (row0 + row) * kC, tid, r);
                }
            }
        }
        // branch 1 rewrites A (its readers passed the barrier above) and T (read by the same threads)
    }
}

template <int RB, int F>
__global__ __launch_bounds__(256) void k_att_pairs_bf(const PairParams p) {
    using TL = Tile<RB, F>;
    constexpr int ROWS = TL::ROWS;
    __shared__ __attribute__((aligned(16))) unsigned char smem[TL::ABUF_BYTES + TL::T_BYTES];
    uint16_t *A = reinterpret_cast<uint16_t *>(smem);
    float *T = reinterpret_cast<float *>(smem + TL::ABUF_BYTES);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int64_t P = *p.n_pairs;
    if (P < 0 || P > p.cap) P = p.cap;
    const int64_t n_tiles = (P + ROWS - 1) / ROWS;
    f32x4 acc[RB][2];
    const uint4 *wd2 = reinterpret_cast<const uint4 *>(p.wpd2), *wc0 = reinterpret_cast<const uint4 *>(p.wpc0e);
    BPair<F> bf;
    if ((int64_t)blockIdx.x < n_tiles) ring_prime<F>(bf, wd2, wave, lane);

    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t pr0 = tile * ROWS;
#pragma unroll
        for (int c0 = 0; c0 < ROWS; c0 += 32) {
            const int row = c0 + (tid >> 3);
            if (row < ROWS) {
                float dx = 0.f, dy = 0.f;
                if (pr0 + row < P) {
                    const float2 a = reinterpret_cast<const float2 *>(p.agt_ctrs)[p.hi[pr0 + row]];
                    const float2 c = reinterpret_cast<const float2 *>(p.ctx_ctrs)[p.wi[pr0 + row]];
                    dx = a.x - c.x; dy = a.y - c.y;
                }
                lin2_relu_split<F>(A, TL::PLANE, row, tid, dx, dy, p.wd0, p.bd0);
            }
        }
        lds_barrier();
        acc_zero<RB>(acc);
        gemm_pass<RB, F>(A, wd2, wc0, bf, wave, lane, acc);
        acc_store<RB>(T, acc, lane, wave);
        lds_barrier();   // all waves done reading A; T complete
#pragma unroll
        for (int c0 = 0; c0 < ROWS; c0 += 32) {
            const int row = c0 + (tid >> 3);
            if (row < ROWS) {
                RowVals r = row_load(T + c0 * kLDA, tid);
                row_gn(r, tid, p.gd, p.btd, p.eps);
                row_relu(r);
                row_split_store<F>(A, TL::PLANE, row, tid, r);
            }
        }
        lds_barrier();
        acc_zero<RB>(acc);
        gemm_pass<RB, F>(A, wc0, wd2, bf, wave, lane, acc);   // prefetches the next tile's first fragments
        acc_store<RB>(T, acc, lane, wave);   // T's readers (previous row phase) passed the barrier above
        lds_barrier();
#pragma unroll
        for (int c0 = 0; c0 < ROWS; c0 += 32) {
            const int row = c0 + (tid >> 3);
            if (row < ROWS) {
                const int64_t pr = pr0 + row;
                const bool live = pr < P;
                RowVals r = row_load(T + c0 * kLDA, tid);
                if (live) {
                    row_add_global(r, p.U + (int64_t)p.hi[pr] * kC, tid);
                    row_add_global(r, p.V + (int64_t)p.wi[pr] * kC, tid);
                }
                row_gn(r, tid, p.gc, p.btc, p.eps);
                row_relu(r);
                if (live) row_store_global(p.m + pr * kC, tid, r);
            }
        }
        lds_barrier();   // next tile rewrites A (read by the last gemm) and T (read just above)
    }
}

// ------------------------------------------------------------ dispatch -----
// CUs of the current device, asked per call: the library keeps no state, not even a cache (the query is a
// table lookup in the runtime, ~0.1 us, against launches of >= 5 us).
static int cu_count() {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
        return v;
    return 256;
}

// 16-row blocks per tile.  Tiles of <= 32 rows (<= 128 VGPRs, <= 63 KB LDS) and, in the two-plane / one-plane modes,
// 48-row tiles built with the one-set weight ring run two workgroups per CU, taller ones one; the generic rule picks
// the height with the fewest rounds x height, the taller on ties (weight reuse: every tile streams the full weight
// set through its CU's L1 once per relation).
static int pick_rb(int64_t n_rows, int fmt, bool lane_conv = false) {
    const int64_t n_sub = (n_rows + 15) / 16;
    auto slots = [&](int rb) { return (int64_t)cu_count() * (fmt != 0 && (rb <= 2 || rb == LGCN_RB_ONE) ? 2 : 1); };
    auto tiles = [&](int rb) { return (n_sub + rb - 1) / rb; };
    // LaneConv with several forwards in flight: once 32-row tiles would put two workgroups on some CUs (more tiles
    // than CUs), 48-row tiles are the better unit -- they still run two per CU (one-set weight ring, 126 VGPRs,
    // 70 KB LDS), a whole launch fits the chip TWICE (216 tiles at S2: two streams' layers side by side instead of
    // one and a third), and a row costs 1.8 KB instead of 2.5 KB through the CU's L1 per relation.  Measured at S2:
    // 116 k vs 108 k scenes/s with four forwards in flight, 53.1 k vs 54.1 k with one.
    if (lane_conv && LGCN_RB_ONE == 3 && fmt != 0 && tiles(2) > cu_count() && tiles(3) <= slots(3)) return 3;
    int best = 1;
    int64_t best_cost = -1;
    for (int rb = 1; rb <= 4; ++rb) {
        const int64_t cost = ((tiles(rb) + slots(rb) - 1) / slots(rb)) * rb;
        if (best_cost < 0 || cost <= best_cost) { best_cost = cost; best = rb; }
    }
    return best;
}

// Tuning knobs of the diagnostic builds (-DLGCN_TUNING: make stamps / ablate), read from the environment once:
// LGCN_RB / LGCN_RB_LC force the tile height, LGCN_RING=3 the deep weight ring, LGCN_EXP_PAD_LDS pads the LaneConv
// workgroups with dynamic LDS.  The shipped library reads no environment variable.
static int env_int(const char *name, int dflt) {
#ifdef LGCN_TUNING
    const char *v = std::getenv(name);
    return v && *v ? std::atoi(v) : dflt;
#else
    (void)name;
    return dflt;
#endif
}

template <int F, bool DEEP>
static void launch_agg(const lgcn_agg_mlp_t &p, int rb, bool lane_conv, hipStream_t st) {
    const int rows = 16 * rb;
    const int n_tiles = (int)((p.n_rows + rows - 1) / rows);
    static const int pad_lds = env_int("LGCN_EXP_PAD_LDS", 0);     // experiment: extra dynamic LDS per workgroup
#define LGCN_AGG(RB_)                                                                                                \
    if (lane_conv) hipLaunchKernelGGL((k_agg_mlp_bf<RB_, F, 1, DEEP>), dim3(n_tiles), dim3(512), pad_lds, st, p, n_tiles); \
    else hipLaunchKernelGGL((k_agg_mlp_bf<RB_, F, 0, DEEP>), dim3(n_tiles), dim3(512), 0, st, p, n_tiles)
    switch (rb) {
        case 1: LGCN_AGG(1); break;
        case 2: LGCN_AGG(2); break;
        case 3: LGCN_AGG(3); break;
        default: LGCN_AGG(4); break;
    }
#undef LGCN_AGG
}

// LGCN_MMA_* -> format id of Fmt<>
static int fmt_of(int mma) { return mma == LGCN_MMA_BF16X3 ? 0 : mma == LGCN_MMA_F16X2 ? 1 : 2; }

int agg_mlp_bf(const lgcn_agg_mlp_t &p, bool lane_conv, hipStream_t st) {
    static const int force_rb = env_int("LGCN_RB", 0), force_rb_lc = env_int("LGCN_RB_LC", 0), ring = env_int("LGCN_RING", 1);
    int rb = p.tile_rb;
    if (rb < 0 || rb > 4) return LGCN_EINVAL;
    if (rb == 0 && lane_conv && force_rb_lc >= 1 && force_rb_lc <= 4) rb = force_rb_lc;
    if (rb == 0) rb = force_rb >= 1 && force_rb <= 4 ? force_rb : pick_rb(p.n_rows, fmt_of(p.mma), lane_conv);
#ifdef LGCN_TUNING      // the deep weight ring (measured: no gain, also for the small row blocks) exists in the tuning builds only
    if (ring >= 3) {
        switch (fmt_of(p.mma)) {
            case 0: launch_agg<0, true>(p, rb, lane_conv, st); break;
            case 1: launch_agg<1, true>(p, rb, lane_conv, st); break;
            default: launch_agg<2, true>(p, rb, lane_conv, st); break;
        }
        return launch_status();
    }
#endif
    (void)ring;
    switch (fmt_of(p.mma)) {
        case 0: launch_agg<0, false>(p, rb, lane_conv, st); break;
        case 1: launch_agg<1, false>(p, rb, lane_conv, st); break;
        default: launch_agg<2, false>(p, rb, lane_conv, st); break;
    }
    return launch_status();
}

int agg_mlp_pair_bf(const lgcn_agg_mlp_t &a, const lgcn_agg_mlp_t &b, hipStream_t st) {
    // one tile height for both problems: the one picked for the larger of the two
    const int rb = pick_rb(a.n_rows > b.n_rows ? a.n_rows : b.n_rows, fmt_of(a.mma));
    const int rows = 16 * rb;
    const int ta = (int)((a.n_rows + rows - 1) / rows), tb = (int)((b.n_rows + rows - 1) / rows);
#define LGCN_AGG2(RB_, F_) hipLaunchKernelGGL((k_agg_mlp_bf2<RB_, F_>), dim3(ta + tb), dim3(512), 0, st, a, b, ta, tb)
#define LGCN_AGG2_RB(F_) switch (rb) { case 1: LGCN_AGG2(1, F_); break; case 2: LGCN_AGG2(2, F_); break; case 3: LGCN_AGG2(3, F_); break; default: LGCN_AGG2(4, F_); }
    switch (fmt_of(a.mma)) {
        case 0: LGCN_AGG2_RB(0); break;
        case 1: LGCN_AGG2_RB(1); break;
        default: LGCN_AGG2_RB(2); break;
    }
#undef LGCN_AGG2_RB
#undef LGCN_AGG2
    return launch_status();
}

int agg_mlp_multi_bf(const lgcn_agg_mlp_t *const *ps, int n, hipStream_t st) {
    // one tile height for all problems: the one picked for the largest
    int64_t big = 0;
    for (int i = 0; i < n; ++i) big = ps[i]->n_rows > big ? ps[i]->n_rows : big;
    if (ps[0]->tile_rb < 0 || ps[0]->tile_rb > 4) return LGCN_EINVAL;
    const int rb = ps[0]->tile_rb ? ps[0]->tile_rb : pick_rb(big, fmt_of(ps[0]->mma));      // one tile height for all (the caller's, or picked)
    const int rows = 16 * rb;
    MultiArgs m{};
    m.n = n;
    int total = 0;
    for (int i = 0; i < n; ++i) {
        m.p[i] = *ps[i];
        m.tiles[i] = (int)((ps[i]->n_rows + rows - 1) / rows);
        total += m.tiles[i];
    }
#define LGCN_AGGN(RB_, F_) hipLaunchKernelGGL((k_agg_mlp_bfn<RB_, F_>), dim3(total), dim3(512), 0, st, m)
#define LGCN_AGGN_RB(F_) switch (rb) { case 1: LGCN_AGGN(1, F_); break; case 2: LGCN_AGGN(2, F_); break; case 3: LGCN_AGGN(3, F_); break; default: LGCN_AGGN(4, F_); }
    switch (fmt_of(ps[0]->mma)) {
        case 0: LGCN_AGGN_RB(0); break;
        case 1: LGCN_AGGN_RB(1); break;
        default: LGCN_AGGN_RB(2); break;
    }
#undef LGCN_AGGN_RB
#undef LGCN_AGGN
    return launch_status();
}

int mapnet_input_bf(const InputParams &p, int mma, hipStream_t st) {
    const int rb = pick_rb(p.n_rows, fmt_of(mma));
    const int n_tiles = (int)((p.n_rows + 16 * rb - 1) / (16 * rb));
#define LGCN_IN(RB_, F_) hipLaunchKernelGGL((k_mapnet_input_bf<RB_, F_>), dim3(n_tiles), dim3(256), 0, st, p, n_tiles)
#define LGCN_IN_RB(F_) switch (rb) { case 1: LGCN_IN(1, F_); break; case 2: LGCN_IN(2, F_); break; case 3: LGCN_IN(3, F_); break; default: LGCN_IN(4, F_); }
    switch (fmt_of(mma)) {
        case 0: LGCN_IN_RB(0); break;
        case 1: LGCN_IN_RB(1); break;
        default: LGCN_IN_RB(2); break;
    }
#undef LGCN_IN_RB
#undef LGCN_IN
    return launch_status();
}

int att_pairs_bf(const PairParams &p, int mma, hipStream_t st) {
    // 32-pair tiles (RB = 2): ~43 KB of LDS, so 3 workgroups share a CU and cover each other's
    // row phases (this kernel has no separate gather waves)
    const int64_t tiles = (p.cap + 31) / 32;
    const int64_t slots = (int64_t)cu_count() * 3;
    const unsigned grid = (unsigned)(tiles < slots ? tiles : slots);
    switch (fmt_of(mma)) {
        case 0: hipLaunchKernelGGL((k_att_pairs_bf<2, 0>), dim3(grid), dim3(256), 0, st, p); break;
        case 1: hipLaunchKernelGGL((k_att_pairs_bf<2, 1>), dim3(grid), dim3(256), 0, st, p); break;
        default: hipLaunchKernelGGL((k_att_pairs_bf<2, 2>), dim3(grid), dim3(256), 0, st, p); break;
    }
    return launch_status();
}

int pack_weight_bf(const float *W, int ld, int mma, int transpose, void *out, hipStream_t st) {
    uint16_t *o = reinterpret_cast<uint16_t *>(out);
    switch (fmt_of(mma)) {
        case 0: hipLaunchKernelGGL((k_pack_weight_bf<0>), dim3(8), dim3(256), 0, st, W, ld, o, transpose); break;
        case 1: hipLaunchKernelGGL((k_pack_weight_bf<1>), dim3(8), dim3(256), 0, st, W, ld, o, transpose); break;
        default: hipLaunchKernelGGL((k_pack_weight_bf<2>), dim3(8), dim3(256), 0, st, W, ld, o, transpose); break;
    }
    return launch_status();
}

int pack_weight_batch_bf(const lgcn_pack_job_t *jobs, int n_jobs, int mma, hipStream_t st) {
    switch (fmt_of(mma)) {
        case 0: hipLaunchKernelGGL((k_pack_weight_batch_bf<0>), dim3(8, n_jobs), dim3(256), 0, st, jobs); break;
        case 1: hipLaunchKernelGGL((k_pack_weight_batch_bf<1>), dim3(8, n_jobs), dim3(256), 0, st, jobs); break;
        default: hipLaunchKernelGGL((k_pack_weight_batch_bf<2>), dim3(8, n_jobs), dim3(256), 0, st, jobs); break;
    }
    return launch_status();
}

}  // namespace lgcn
