// Att.forward (reference lanegcn.py:691-709) for a tile of TARGET rows in one launch: the per-target query path,
// the per-pair MLP, the segment sum over the target's pairs and the node epilogue.  The pair rows m_p never leave
// the CU (the three-launch version writes [cap,128] fp32 to HBM and reads it back: 0.3-1.6 GB of buffer per call on
// large batches), and what was three launches per Att layer (U/V, pairs, tail) is one plus the per-context V GEMM.
// Measured at S2: 1.2-1.6x SLOWER than the three wide launches (a workgroup walks ~7-20 dependent 32-row GEMM
// passes with its weights streamed from L2 each time, at one or two workgroups per CU): this is the memory-lean
// option (ops.set_att_impl("fused")), not the default.
//
//   U[t]  = ReLU(GN_q(W_q a[t])) W_c0[:,128:256]^T                               per target, 2 GEMMs
//   e_p   = ReLU(GN_d(W_d2 ReLU(W_d0 (c_agt[h_p] - c_ctx[w_p]) + b_d0)))          per pair,   1 GEMM
//   m_p   = ReLU(GN_c(W_c0[:,0:128] e_p + U[h_p] + V[w_p]))                       per pair,   1 GEMM (V: own launch)
//   S[t]  = sum_{p: h_p = t} m_p                                                 pairs are sorted by h (row-major
//                                                                                nonzero per scene): contiguous
//   a'[t] = ReLU(GN_l(W_lin ReLU(GN_n(W_agt a[t] + W_c1 S[t]))) + a[t])           per target, 3 GEMMs
//
// One workgroup = TT consecutive targets (8 / 16 / 32: the host picks it so that the tiles fill the chip) and all
// their pairs, 32 at a time.  4 waves; every GEMM is the 32 x 128 x 128 pass of lgcn_mma_bf.hpp (A planes in LDS,
// wave w owns 32 output channels and streams its weight slice L2 -> VGPR, the next GEMM's first K-step prefetched
// behind the current one); row phases with 8 threads per row.  The segment sum is done by the thread group that
// owns the target row, in pair order, in registers: no atomics, bitwise repeatable.
#include "lgcn_common.hpp"
#include "lgcn_tile.hpp"
#include "lgcn_mma_bf.hpp"

namespace lgcn {

struct AttFusedParams {
    const float *agts;                 // [T,128]
    int64_t n_agt;
    const float *agt_ctrs, *ctx_ctrs;  // [T,2], [S,2]
    const int32_t *hi, *wi, *rowptr;   // pairs sorted by hi; rowptr[t] = first pair with hi >= t, rowptr[T] = P
    int64_t cap;
    const float *wpq, *gq, *bq, *wpc0q;                               // query -> U
    const float *wd0, *bd0, *wpd2, *gd, *btd, *wpc0e, *V, *gc, *btc;   // per-pair MLP
    const float *wpagt, *wpc1, *gn, *bn, *wplin, *gl, *bl;             // epilogue
    float eps;
    float *out;
    int tt;                            // targets per workgroup: 8, 16 or 32
};

template <int F>
__global__ __launch_bounds__(256) void k_att_fused(const AttFusedParams p) {
    using TL = Tile<2, F>;
    __shared__ __attribute__((aligned(16))) unsigned char smem[TL::ABUF_BYTES + 2 * TL::T_BYTES];
    uint16_t *A = reinterpret_cast<uint16_t *>(smem);
    float *T = reinterpret_cast<float *>(smem + TL::ABUF_BYTES);
    float *U = reinterpret_cast<float *>(smem + TL::ABUF_BYTES + TL::T_BYTES);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row = tid >> 3;
    const int TT = p.tt;
    const int64_t t0 = (int64_t)blockIdx.x * TT;
    const int64_t tn = t0 + row;
    const bool own = row < TT && tn < p.n_agt;        // this thread group owns target row `row` of the tile
    auto clampp = [&](int v) { return v < 0 ? 0 : ((int64_t)v > p.cap ? (int)p.cap : v); };
    const int64_t tlast = t0 + TT < p.n_agt ? t0 + TT : p.n_agt;
    const int p_begin = clampp(p.rowptr[t0]), p_end = clampp(p.rowptr[tlast]);
    const int rp0 = own ? clampp(p.rowptr[tn]) : 0, rp1 = own ? clampp(p.rowptr[tn + 1]) : 0;

    const uint4 *wq = reinterpret_cast<const uint4 *>(p.wpq), *wc0q = reinterpret_cast<const uint4 *>(p.wpc0q);
    const uint4 *wd2 = reinterpret_cast<const uint4 *>(p.wpd2), *wc0e = reinterpret_cast<const uint4 *>(p.wpc0e);
    const uint4 *wagt = reinterpret_cast<const uint4 *>(p.wpagt), *wc1 = reinterpret_cast<const uint4 *>(p.wpc1);
    const uint4 *wlin = reinterpret_cast<const uint4 *>(p.wplin);
    BPair<F> bf;
    ring_prime<F>(bf, wq, wave, lane);
    f32x4 acc[2][2];

    // ---- target rows: a[t] (kept: second operand of the epilogue and its residual), U[t]
    RowVals arow;
    {
        const float *rp_ = p.agts + (own ? tn : 0) * kC + 4 * (tid & 7);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float4 v = *reinterpret_cast<const float4 *>(rp_ + 32 * j);
            arow.v[j] = own ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    row_split_store<F>(A, TL::PLANE, row, tid, arow);
    __syncthreads();
    acc_zero<2>(acc);
    gemm_pass<2, F>(A, wq, wc0q, bf, wave, lane, acc);
    acc_store<2>(T, acc, lane, wave);
    __syncthreads();
    {
        RowVals r = row_load(T, tid);
        row_gn(r, tid, p.gq, p.bq, p.eps);
        row_relu(r);
        row_split_store<F>(A, TL::PLANE, row, tid, r);
    }
    __syncthreads();
    acc_zero<2>(acc);
    gemm_pass<2, F>(A, wc0q, p_begin < p_end ? wd2 : wc1, bf, wave, lane, acc);
    acc_store<2>(U, acc, lane, wave);
    // (U is read after the barriers of the first pair chunk)

    // ---- pairs, 32 at a time; S = the owned target row's running segment sum
    RowVals S;
#pragma unroll
    for (int j = 0; j < 4; ++j) S.v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int c0 = p_begin; c0 < p_end; c0 += 32) {
        const int pr = c0 + row;
        const bool live = pr < p_end;
        int hl = 0, w = 0;
        float dx = 0.f, dy = 0.f;
        if (live) {
            const int h = p.hi[pr];
            w = p.wi[pr];
            hl = h - (int)t0;
            const float2 a = reinterpret_cast<const float2 *>(p.agt_ctrs)[h];
            const float2 c = reinterpret_cast<const float2 *>(p.ctx_ctrs)[w];
            dx = a.x - c.x;
            dy = a.y - c.y;
        }
        hl = hl < 0 ? 0 : (hl > 31 ? 31 : hl);          // never index outside the tile, whatever the index says
        RowVals vrow;                                    // V[w]: requested here, used two GEMMs later
        {
            const float *vp = p.V + (int64_t)w * kC + 4 * (tid & 7);
#pragma unroll
            for (int j = 0; j < 4; ++j) vrow.v[j] = *reinterpret_cast<const float4 *>(vp + 32 * j);
        }
        lin2_relu_split<F>(A, TL::PLANE, row, tid, dx, dy, p.wd0, p.bd0);
        __syncthreads();
        acc_zero<2>(acc);
        gemm_pass<2, F>(A, wd2, wc0e, bf, wave, lane, acc);
        acc_store<2>(T, acc, lane, wave);
        __syncthreads();
        {
            RowVals r = row_load(T, tid);
            row_gn(r, tid, p.gd, p.btd, p.eps);
            row_relu(r);
            row_split_store<F>(A, TL::PLANE, row, tid, r);
        }
        __syncthreads();
        acc_zero<2>(acc);
        gemm_pass<2, F>(A, wc0e, c0 + 32 < p_end ? wd2 : wc1, bf, wave, lane, acc);
        acc_store<2>(T, acc, lane, wave);
        __syncthreads();
        {
            RowVals r = row_load(T, tid);
            row_add(r, row_load(U + (hl - row) * kLDA, tid));      // row_load adds (tid >> 3) rows itself
            row_add(r, vrow);
            row_gn(r, tid, p.gc, p.btc, p.eps);
            row_relu(r);
            if (!live) {
#pragma unroll
                for (int j = 0; j < 4; ++j) r.v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            row_store_lds(T, tid, r);                               // m_p, same positions this thread just read
        }
        __syncthreads();
        if (own) {      // this target's pairs inside the chunk, in pair order
            const int b = rp0 > c0 ? rp0 : c0, e = rp1 < c0 + 32 ? rp1 : c0 + 32;
            for (int q = b; q < e; ++q) row_add(S, row_load(T + (q - c0 - row) * kLDA, tid));
        }
        // (T is rewritten behind the next chunk's first barrier; A by its first row phase: read before this one's last)
    }

    // ---- epilogue: W_c1 S + W_agt a -> GN -> ReLU -> W_lin -> GN -> + a -> ReLU
    __syncthreads();
    row_split_store<F>(A, TL::PLANE, row, tid, S);
    __syncthreads();
    acc_zero<2>(acc);
    gemm_pass<2, F>(A, wc1, wagt, bf, wave, lane, acc);
    __syncthreads();                                  // every wave is done reading the S planes
    row_split_store<F>(A, TL::PLANE, row, tid, arow);
    __syncthreads();
    gemm_pass<2, F>(A, wagt, wlin, bf, wave, lane, acc);
    acc_store<2>(T, acc, lane, wave);
    __syncthreads();
    {
        RowVals r = row_load(T, tid);
        row_gn(r, tid, p.gn, p.bn, p.eps);
        row_relu(r);
        row_split_store<F>(A, TL::PLANE, row, tid, r);
    }
    __syncthreads();
    acc_zero<2>(acc);
    gemm_pass<2, F>(A, wlin, nullptr, bf, wave, lane, acc);
    acc_store<2>(T, acc, lane, wave);
    __syncthreads();
    if (own) {
        RowVals r = row_load(T, tid);
        row_gn(r, tid, p.gl, p.bl, p.eps);
        row_add(r, arow);
        row_relu(r);
        row_store_global(p.out + tn * kC, tid, r);
    }
}


#ifdef LGCN_STAMPS
__device__ unsigned long long *g_att_stamps = nullptr;     // diagnostic build only (tools/stamps_att.py)
#endif

// --------------------------------------------------- weight-stationary pair MLP -----
// lgcn_att_pairs_ws: the per-pair MLP of Att.forward (lanegcn.py:691-700) with both 128 x 128 weights held in
// REGISTERS for the whole launch.  k_att_pairs_bf streams 128 KB of weight fragments L2 -> VGPR per 32-pair tile
// (32 vector-memory instructions per wave and tile, each ~60-100 cycles of issue): at S2 that stream, not the
// matrix cores, is what its 12-33 us are.  Here a persistent workgroup of 8 waves takes 64-pair tiles; wave w owns
// output channels 16 w .. 16 w + 15 of BOTH GEMMs (its slice of W_d2 and W_c0e: 2 x 8 KB = 64 VGPRs in f16x2), the A
// operand planes and the fp32 tile live in LDS (<= 80 KB: two workgroups per CU cover each other's row phases), the
// small per-channel parameters (W_d0, b_d0, the two GroupNorms) are staged in LDS once.  Vector-memory instructions
// per wave and tile: 8 (the U / V rows, which land directly in the second GEMM's accumulators) + the output rows.
//
// seg = 0: m[p] = m_p for every pair (what lgcn_att_pairs writes).
// seg = 16: pairs are sorted by target; within every 16-aligned group of pair rows the rows of one target are summed
//   in pair order and the sum is written at the row of the piece's FIRST pair; the other rows are not written.
//   The tail then adds, per target with pairs [a, b), the rows {a} U {16 j : a < 16 j < b} (relation mode
//   LGCN_REL_RANGE16 of lgcn_agg_mlp): 12 x fewer rows through HBM and through the tail's CUs for A2A at S2.
#ifndef LGCN_WS_WAVES      // diagnostic builds only (make relucnd_nolimit / spill): another register budget for this kernel
#define LGCN_WS_WAVES 4
#endif
template <int F>
#ifdef LGCN_WS_NUMVGPR
__global__ __launch_bounds__(512) __attribute__((amdgpu_num_vgpr(LGCN_WS_NUMVGPR)))
#else
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(F == 0 ? 2 : LGCN_WS_WAVES)))
#endif
void k_att_pairs_ws(const PairParams p, const int seg) {
    constexpr int RB = 4, ROWS = 64, NP = Fmt<F>::NP;
#ifdef LGCN_STAMPS
    unsigned long long *sbuf = (g_att_stamps && (threadIdx.x & 63) == 0 && ((threadIdx.x >> 6) == 0 || (threadIdx.x >> 6) == 7))
                                   ? g_att_stamps + ((int64_t)blockIdx.x * 2 + ((threadIdx.x >> 6) == 7)) * 32 : nullptr;
    int sidx = 0;
#define AT_STAMP() do { if (sbuf && sidx < 32) sbuf[sidx] = stamp(); ++sidx; } while (0)
#else
#define AT_STAMP() do { } while (0)
#endif
    AT_STAMP();   // 0 start
    using TL = Tile<RB, F>;
    __shared__ __attribute__((aligned(16))) unsigned char smem[TL::ABUF_BYTES + TL::T_BYTES];
    __shared__ __attribute__((aligned(16))) float s_par[7 * kC];      // wd0 [128][2] | bd0 | gd | btd | gc | btc
    __shared__ __attribute__((aligned(16))) int s_hi[ROWS], s_wi[ROWS];
    uint16_t *A = reinterpret_cast<uint16_t *>(smem);
    float *T = reinterpret_cast<float *>(smem + TL::ABUF_BYTES);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int64_t P = *p.n_pairs;
    if (P < 0 || P > p.cap) P = p.cap;
    const int64_t n_tiles = (P + ROWS - 1) / ROWS;
    if ((int64_t)blockIdx.x >= n_tiles) return;

    const int row = tid >> 3;
    // pair indices and centre offsets of the first tile (two dependent round trips: requested first, the weight
    // slices travel beside them); later tiles: requested one tile ahead
    int64_t tile = blockIdx.x;
    int hi_c = -1, wi_c = 0;
    float dx = 0.f, dy = 0.f;
    auto fetch_idx = [&](int64_t tl, int &h, int &w) {
        const int64_t pr = tl * ROWS + row;
        const bool live = tl < n_tiles && pr < P;
        h = live ? p.hi[pr] : -1;
        w = live ? p.wi[pr] : 0;
    };
    auto fetch_d = [&](int h, int w, float &x, float &y) {
        x = y = 0.f;
        if (h >= 0) {
            const float2 a = reinterpret_cast<const float2 *>(p.agt_ctrs)[h];
            const float2 c = reinterpret_cast<const float2 *>(p.ctx_ctrs)[w];
            x = a.x - c.x; y = a.y - c.y;
        }
    };
    fetch_idx(tile, hi_c, wi_c);

    // this wave's slices of the two weights: [plane][K-step], channels 16 wave .. + 15.  W_d2 first: the first GEMM
    // waits for it alone, W_c0e lands under that GEMM
    uint4 w1[NP][4], w2[NP][4];
    const uint4 *B1 = reinterpret_cast<const uint4 *>(p.wpd2), *B2 = reinterpret_cast<const uint4 *>(p.wpc0e);
#pragma unroll
    for (int pl = 0; pl < NP; ++pl)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            w1[pl][ks] = B1[((((pl * 4 + (wave >> 1)) * 4 + ks) * 2 + (wave & 1)) << 6) + lane];
    if (tid < 2 * kC) s_par[tid] = p.wd0[tid];
    else if (tid < 3 * kC) s_par[tid] = p.bd0[tid - 2 * kC];
    else if (tid < 4 * kC) s_par[tid] = p.gd[tid - 3 * kC];
    if (tid < kC) {
        s_par[4 * kC + tid] = p.btd[tid];
        s_par[5 * kC + tid] = p.gc[tid];
        s_par[6 * kC + tid] = p.btc[tid];
    }
    fetch_d(hi_c, wi_c, dx, dy);
#pragma unroll
    for (int pl = 0; pl < NP; ++pl)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            w2[pl][ks] = B2[((((pl * 4 + (wave >> 1)) * 4 + ks) * 2 + (wave & 1)) << 6) + lane];
    const float *l_wd0 = s_par, *l_bd0 = s_par + 2 * kC, *l_gd = s_par + 3 * kC, *l_btd = s_par + 4 * kC;
    const float *l_gc = s_par + 5 * kC, *l_btc = s_par + 6 * kC;

    const uint16_t *arow = A + (lane & 15) * kLDB + 8 * (lane >> 4);
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 acc[RB];
    // acc[rb] += W-slice^T x A^T (swapped operands: a lane ends up with 4 consecutive channels of row lane & 15)
    auto gemm = [&](const uint4 (&w)[NP][4]) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
            for (int r2 = 0; r2 < RB; r2 += 2) {      // two sub-blocks' fragments at a time (register budget)
                uint4 a[NP][2];
#pragma unroll
                for (int pl = 0; pl < NP; ++pl)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        a[pl][j] = *reinterpret_cast<const uint4 *>(arow + pl * TL::PLANE + (r2 + j) * 16 * kLDB + 32 * ks);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    f32x4 c = acc[r2 + j];
#pragma unroll
                    for (int q = 0; q < Fmt<F>::NPROD; ++q)      // smallest terms first
                        c = Fmt<F>::mfma(w[Fmt<F>::PB[q]][ks], a[Fmt<F>::PA[q]][j], c);
                    acc[r2 + j] = c;
                }
            }
        }
    };
    auto acc_to_tile = [&]() {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
            *reinterpret_cast<f32x4 *>(T + (16 * rb + (lane & 15)) * kLDA + 16 * wave + 4 * (lane >> 4)) = acc[rb];
    };
    AT_STAMP();   // 1 requests issued, centres landed
    lds_barrier();          // s_par
    AT_STAMP();   // 2

    for (; tile < n_tiles; tile += gridDim.x) {
        const int64_t pr0 = tile * ROWS;
        // the opaque zero keeps the row phases' per-thread addressing (the same for every tile) out of the loop
        // pre-header: hoisted, those values would live in registers / scratch for the whole kernel (128-VGPR budget)
        int opq = 0;
        asm volatile("" : "+v"(opq));
        const int tidv = tid + opq, rowv = tidv >> 3, lanev = tidv & 63;
        // ---- e0 = ReLU(W_d0 d + b_d0) -> A planes; the tile's pair indices -> LDS; next tile's indices requested
        lin2_relu_split<F>(A, TL::PLANE, rowv, tidv, dx, dy, l_wd0, l_bd0);
        if ((tidv & 7) == 0) { s_hi[rowv] = hi_c; s_wi[rowv] = wi_c; }
        int hi_n, wi_n;
        fetch_idx(tile + gridDim.x, hi_n, wi_n);
        lds_barrier();
        AT_STAMP();   // 3 e0 planes ready
        // ---- e1 = A x W_d2
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) acc[rb] = zero4;
        gemm(w1);
        AT_STAMP();   // 4 first GEMM issued
        acc_to_tile();
        // U[hi] + V[wi] of this lane's rows / channels: the second GEMM's accumulators start from them
        // (registers: the U rows travel under the row phase, the V rows are requested behind it)
        const int64_t co = 16 * wave + 4 * (lanev >> 4);
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            const int h = s_hi[16 * rb + (lanev & 15)];
            acc[rb] = *reinterpret_cast<const f32x4 *>(p.U + (int64_t)(h < 0 ? 0 : h) * kC + co);
        }
        lds_barrier();      // T complete; every wave is done reading A
        AT_STAMP();   // 5
        // ---- e = ReLU(GN_d(e1)) -> A planes
        {
            RowVals r = row_load(T, tidv);
            row_gn(r, tidv, l_gd, l_btd, p.eps);
            row_relu(r);
            row_split_store<F>(A, TL::PLANE, rowv, tidv, r);
        }
        fetch_d(hi_n, wi_n, dx, dy);          // next tile's centres (its indices have landed by now)
        {
            f32x4 v[RB];
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const int r = 16 * rb + (lanev & 15);
                const int w = s_hi[r] < 0 ? 0 : s_wi[r];
                v[rb] = *reinterpret_cast<const f32x4 *>(p.V + (int64_t)w * kC + co);
            }
            AT_STAMP();   // 6 row phase done
            lds_barrier();
            AT_STAMP();   // 7
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) acc[rb] = acc[rb] + v[rb];
        }
        AT_STAMP();   // 8 U + V in the accumulators
        // ---- t = U + V + A x W_c0e
        gemm(w2);
        AT_STAMP();   // 9
        acc_to_tile();
        lds_barrier();
        AT_STAMP();   // 10
        // ---- m_p = ReLU(GN_c(t))
        {
            RowVals r = row_load(T, tidv);
            row_gn(r, tidv, l_gc, l_btc, p.eps);
            row_relu(r);
            const int64_t pr = pr0 + rowv;
            if (seg == 0) {
                if (pr < P) row_store_global(p.m + pr * kC, tidv, r);
            } else {
                row_store_lds(T, tidv, r);     // the thread's own 16 floats: no other thread touches them
            }
        }
        AT_STAMP();   // 11 last row phase done
        if (seg != 0) {
            lds_barrier();
            // one thread per (channel, 16-row group): the rows of a target are summed in pair order.  The group's 16
            // target ids (wave-uniform) and the thread's 16 values are fetched up front; the walk itself is
            // registers only, a store per piece.
            const int c = tidv & (kC - 1), g0 = (tidv >> 7) * 16;
            int t[17];
            float x[16];
#pragma unroll
            for (int i = 0; i < 16; i += 4) {
                const int4 q = *reinterpret_cast<const int4 *>(s_hi + g0 + i);
                t[i] = q.x; t[i + 1] = q.y; t[i + 2] = q.z; t[i + 3] = q.w;
            }
            t[16] = -2;                                    // closes the last piece of the group
#pragma unroll
            for (int i = 0; i < 16; ++i) x[i] = T[(g0 + i) * kLDA + c];
            float sum = 0.f;
            int first = 0;
            float *mrow = p.m + (pr0 + g0) * kC + c;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                sum += x[i];
                if (t[i + 1] != t[i]) {                    // wave-uniform
                    if (t[i] >= 0) mrow[first * kC] = sum;
                    sum = 0.f;
                    first = i + 1;
                }
            }
        }
        hi_c = hi_n; wi_c = wi_n;
        AT_STAMP();   // 12 pieces written
        lds_barrier();      // the next tile rewrites A, T and the index words
        AT_STAMP();   // 13 tile done
    }
}

}  // namespace lgcn

using namespace lgcn;

extern "C" int lgcn_att_fused(const lgcn_att_fused_t *ph, void *stream) {
    LGCN_CHECK_PTR(ph);
    const lgcn_att_fused_t &q = *ph;
    if (q.mma != LGCN_MMA_BF16X3 && q.mma != LGCN_MMA_F16X2 && q.mma != LGCN_MMA_BF16) return LGCN_ESHAPE;
    if (q.n_agt < 0 || q.cap < 0 || (q.targets_per_block != 4 && q.targets_per_block != 8 && q.targets_per_block != 16 && q.targets_per_block != 32))
        return LGCN_EINVAL;
    if (q.n_agt == 0) return LGCN_OK;
    if (q.n_agt > 0x7fffffff || q.cap > 0x7ffffff0) return LGCN_ESHAPE;
    const void *al[] = {q.agts, q.wpq, q.gq, q.bq, q.wpc0q, q.wd0, q.bd0, q.wpd2, q.gd, q.btd, q.wpc0e, q.V, q.gc, q.btc,
                        q.wpagt, q.wpc1, q.gn, q.bn, q.wplin, q.gl, q.bl, q.out};
    for (const void *v : al) { LGCN_CHECK_PTR(v); LGCN_CHECK_ALIGN16(v); }
    const void *pl[] = {q.agt_ctrs, q.ctx_ctrs, q.hi, q.wi, q.rowptr};
    for (const void *v : pl) LGCN_CHECK_PTR(v);
    AttFusedParams p{q.agts, q.n_agt, q.agt_ctrs, q.ctx_ctrs, q.hi, q.wi, q.rowptr, q.cap,
                     q.wpq, q.gq, q.bq, q.wpc0q, q.wd0, q.bd0, q.wpd2, q.gd, q.btd, q.wpc0e, q.V, q.gc, q.btc,
                     q.wpagt, q.wpc1, q.gn, q.bn, q.wplin, q.gl, q.bl, q.eps, q.out, q.targets_per_block};
    const unsigned grid = (unsigned)((q.n_agt + q.targets_per_block - 1) / q.targets_per_block);
    hipStream_t st = (hipStream_t)stream;
    switch (q.mma) {
        case LGCN_MMA_BF16X3: hipLaunchKernelGGL((k_att_fused<0>), dim3(grid), dim3(256), 0, st, p); break;
        case LGCN_MMA_F16X2: hipLaunchKernelGGL((k_att_fused<1>), dim3(grid), dim3(256), 0, st, p); break;
        default: hipLaunchKernelGGL((k_att_fused<2>), dim3(grid), dim3(256), 0, st, p); break;
    }
    return launch_status();
}

extern "C" int lgcn_att_pairs_ws(const float *agt_ctrs, const float *ctx_ctrs, const int32_t *hi, const int32_t *wi,
                                 const int32_t *n_pairs, int64_t cap, const float *wd0, const float *bd0,
                                 const float *wpd2, const float *gd, const float *btd, const float *wpc0e,
                                 const float *U, const float *V, const float *gc, const float *btc, float eps, int mma,
                                 int seg, float *m, void *stream) {
    if (mma != LGCN_MMA_BF16X3 && mma != LGCN_MMA_F16X2 && mma != LGCN_MMA_BF16) return LGCN_ESHAPE;
    if (cap < 0 || (seg != 0 && seg != 16)) return LGCN_EINVAL;
    if (cap == 0) return LGCN_OK;
    if (cap > 0x7ffffff0) return LGCN_ESHAPE;
    const void *ptrs[] = {agt_ctrs, ctx_ctrs, hi, wi, n_pairs, wd0, bd0, wpd2, gd, btd, wpc0e, U, V, gc, btc, m};
    for (const void *q : ptrs) LGCN_CHECK_PTR(q);
    const void *al[] = {wd0, bd0, wpd2, gd, btd, wpc0e, U, V, gc, btc, m};
    for (const void *q : al) LGCN_CHECK_ALIGN16(q);
    PairParams p{agt_ctrs, ctx_ctrs, hi, wi, n_pairs, cap, wd0, bd0, wpd2, gd, btd, wpc0e, U, V, gc, btc, eps, m};
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
        cus = 256;
    const int64_t tiles = (cap + 63) / 64;
    const int64_t slots = (int64_t)cus * (mma == LGCN_MMA_BF16X3 ? 1 : 2);
    const unsigned grid = (unsigned)(tiles < slots ? tiles : slots);
    hipStream_t st = (hipStream_t)stream;
    switch (mma) {
        case LGCN_MMA_BF16X3: hipLaunchKernelGGL((k_att_pairs_ws<0>), dim3(grid), dim3(512), 0, st, p, seg); break;
        case LGCN_MMA_F16X2: hipLaunchKernelGGL((k_att_pairs_ws<1>), dim3(grid), dim3(512), 0, st, p, seg); break;
        default: hipLaunchKernelGGL((k_att_pairs_ws<2>), dim3(grid), dim3(512), 0, st, p, seg); break;
    }
    return launch_status();
}

#ifdef LGCN_STAMPS
extern "C" void lgcn_debug_att_stamps(void *buf) {
    unsigned long long *p = reinterpret_cast<unsigned long long *>(buf);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(lgcn::g_att_stamps), &p, sizeof(p));
}
#endif
