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
