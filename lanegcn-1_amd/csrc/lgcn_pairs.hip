// lgcn_att_pairs_wi: the per-pair MLP of Att.forward (reference lanegcn.py:691-700) with WAVE-INDEPENDENT 16-pair
// blocks.
//
//   e_p = ReLU(GN_d(W_d2 ReLU(W_d0 (c_agt[h_p] - c_ctx[w_p]) + b_d0)))
//   m_p = ReLU(GN_c(W_c0[:, 0:128] e_p + U[h_p] + V[w_p]))
//
// lgcn_att_pairs_ws keeps both 128 x 128 weights in registers, split over the 8 waves of a workgroup, so every wave
// reads the whole 64-row operand tile from LDS for each GEMM and the tile passes through seven workgroup barriers
// (e0 planes -> GEMM -> fp32 tile -> GroupNorm rows -> planes -> GEMM -> tile -> rows -> piece sums): its phases add
// up (stamps, round 2: ~19 k cycles per 64-pair tile against 3 k of MFMA work) instead of overlapping.
//
// Here ONE wave owns 16 pair rows from the centre offsets to the output and never meets another wave after the
// prologue:
//   * both weights live in LDS for the whole launch (2 x 64 KB of fp16 planes in the f16x2 mode: one workgroup of 16
//     waves per CU), pre-arranged as MFMA fragments so that a fragment is one conflict-free ds_read_b128 per lane;
//   * the GEMMs run "swapped" (D^T = W X^T: the weight is the A operand, the pair rows the B operand), so a lane ends up
//     with 4 consecutive output channels of row lane & 15 for each of the 8 channel blocks -- 32 channels of ONE row;
//   * with the K index of both weights permuted at pack time (lgcn_pack_weight_kperm: K-step s, lane group g, slot j
//     <-> channel 32 s + 16 (j >> 2) + 4 g + (j & 3)), that accumulator layout IS the next GEMM's operand layout: the
//     GroupNorm, the ReLU and the split into fp16 planes happen in registers, the row statistics meet through two
//     ds_bpermute steps (lanes r, r + 16, r + 32, r + 48 hold one row), and nothing of a row ever goes to LDS;
//   * U[h] / V[w] rows are loaded straight into the accumulator layout (16 bytes per lane, 64-byte row segments);
//   * seg = 16: the rows of one target inside the block are summed by a segmented DPP scan over the 16 lanes of a row
//     group (pairs are sorted by target) and only each piece's first row is written -- the same contract as
//     lgcn_att_pairs_ws(seg = 16), read by the tail as an LGCN_REL_RANGE16 relation.
// With 16 waves per CU in different phases the matrix pipe, the VALU and the LDS reads of different blocks overlap; the
// LDS traffic per 16-pair block is 2 x 64 KB of weight fragments (256 B/clk/CU: 2 k cycles per 64 pairs, below the 3 k
// of its MFMAs).
//
// f16x2 (2 fp16 planes, 3 products) and bf16 (1 plane) only: three bf16 planes of two weights (192 KB) do not fit the
// LDS; bf16x3 and f32 keep lgcn_att_pairs_ws / lgcn_att_pairs.
#include "lgcn_common.hpp"
#include "lgcn_tile.hpp"
#include "lgcn_mma_bf.hpp"

// fused multiply-adds allowed in this file (the library is built with -ffp-contract=off for the bit-exact pair search
// of lgcn_index.hip; nothing here is compared bit for bit with ATen): fewer VALU instructions per element, and every
// contraction only removes a rounding
#pragma clang fp contract(fast)

namespace lgcn {

// channel held by (K-step s, lane group g, slot j) of an operand fragment / by accumulator (cb = 2 s + (j >> 2), i = j & 3)
__host__ __device__ constexpr int kperm(int s, int g, int j) { return 32 * s + 16 * (j >> 2) + 4 * g + (j & 3); }

// out[p][cb][s][lane][j] = plane_p(W[16 cb + (lane & 15)][kperm(s, lane >> 4, j)]):  the A operand (weight rows = output
// channels) of v_mfma_f32_16x16x32 for output block cb and K-step s, K index permuted as above
template <int F>
__global__ __launch_bounds__(256) void k_pack_weight_kperm(const float *__restrict__ W, int ld, uint16_t *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;      // one thread per (cb, s, lane)
    if (i >= 8 * 4 * 64) return;
    const int lane = i & 63, s = (i >> 6) & 3, cb = i >> 8;
    const int orow = 16 * cb + (lane & 15), g = lane >> 4;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = W[(int64_t)orow * ld + kperm(s, g, j)];
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
        reinterpret_cast<uint4 *>(out)[(((p * 8 + cb) * 4 + s) << 6) + lane] = make_uint4(q[0], q[1], q[2], q[3]);
    }
}

template <int CTRL>
__device__ __forceinline__ int dpp_int_keep(int old, int x) {      // lanes without a source lane keep `old`
    return __builtin_amdgcn_update_dpp(old, x, CTRL, 0xf, 0xf, false);
}

template <int F, bool SEG>
__global__ __launch_bounds__(1024) void k_att_pairs_wi(const PairParams p) {
    constexpr int NP = Fmt<F>::NP, NPROD = Fmt<F>::NPROD;
    constexpr int WFR = 8 * 4 * 64;                         // uint4 fragments of one plane of one weight
    __shared__ __attribute__((aligned(16))) uint4 s_w[2 * NP * WFR];     // W_d2 planes | W_c0e planes
    __shared__ __attribute__((aligned(16))) float s_par[7 * kC];         // wd0x | wd0y | bd0 | gd | btd | gc | btc
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // scalar: the block index and everything derived from it stay in SGPRs
    const int r = lane & 15, g = lane >> 4;
    int P = *p.n_pairs;                                      // cap <= 0x7ffffff0 (host check): pair indices fit 32 bits
    if (P < 0 || (int64_t)P > p.cap) P = (int)p.cap;
    const int nblk = (P + 15) >> 4;
    if ((int)blockIdx.x >= nblk) return;                    // (uniform per workgroup; grid <= blocks anyway)

    // this wave's first block: its pair indices travel while the weights are copied
    int blk = (int)blockIdx.x + (int)gridDim.x * wave;
    const int stride = (int)gridDim.x * 16;
    auto fetch_idx = [&](int b, int &h, int &w) {
        const bool live = b < nblk && b * 16 + r < P;
        const int32_t *hb = p.hi + (int64_t)b * 16, *wb = p.wi + (int64_t)b * 16;       // scalar bases, lane offset r
        h = live ? hb[r] : -1;
        w = live ? wb[r] : 0;
    };
    int hi_c, wi_c;
    fetch_idx(blk, hi_c, wi_c);
    auto fetch_d = [&](int h, int w, float &x, float &y) {      // centre offset of a pair (0 for a dead row)
        x = y = 0.f;
        if (h >= 0) {
            const float2 a = reinterpret_cast<const float2 *>(p.agt_ctrs)[h];
            const float2 c = reinterpret_cast<const float2 *>(p.ctx_ctrs)[w];
            x = a.x - c.x; y = a.y - c.y;
        }
    };

    float dx, dy;
    {   // weights -> LDS (the packed images are already in fragment order), parameters -> LDS
        const uint4 *B1 = reinterpret_cast<const uint4 *>(p.wpd2), *B2 = reinterpret_cast<const uint4 *>(p.wpc0e);
        uint4 t[2 * NP * WFR / 1024];
#pragma unroll
        for (int i = 0; i < 2 * NP * WFR / 1024; ++i) {
            const int e = tid + 1024 * i;
            t[i] = e < NP * WFR ? B1[e] : B2[e - NP * WFR];
        }
        fetch_d(hi_c, wi_c, dx, dy);      // the first block's centres: requested behind the weight loads, land under the copy
        if (tid < kC) {
            s_par[tid] = p.wd0[2 * tid];
            s_par[kC + tid] = p.wd0[2 * tid + 1];
            s_par[2 * kC + tid] = p.bd0[tid];
            s_par[3 * kC + tid] = p.gd[tid];
            s_par[4 * kC + tid] = p.btd[tid];
            s_par[5 * kC + tid] = p.gc[tid];
            s_par[6 * kC + tid] = p.btc[tid];
        }
#pragma unroll
        for (int i = 0; i < 2 * NP * WFR / 1024; ++i) s_w[tid + 1024 * i] = t[i];
    }
    __syncthreads();        // the only workgroup barrier of the kernel

    const float4 *l_wx = reinterpret_cast<const float4 *>(s_par) + g, *l_wy = reinterpret_cast<const float4 *>(s_par + kC) + g;
    const float4 *l_b0 = reinterpret_cast<const float4 *>(s_par + 2 * kC) + g;
    const float4 *l_gd = reinterpret_cast<const float4 *>(s_par + 3 * kC) + g, *l_bd = reinterpret_cast<const float4 *>(s_par + 4 * kC) + g;
    const float4 *l_gc = reinterpret_cast<const float4 *>(s_par + 5 * kC) + g, *l_bc = reinterpret_cast<const float4 *>(s_par + 6 * kC) + g;
    const uint4 *w1 = s_w + lane, *w2 = s_w + NP * WFR + lane;

    // one K = 128 pass: acc[cb] += W-fragments(cb, s) x X-planes(s); products smallest terms first.  The 32 (K-step,
    // channel block) steps run in order, each step's weight fragments requested two steps ahead (three fragment sets in
    // registers: the scheduler is kept from hoisting all 64 LDS reads to the top, which would cost 256 registers);
    // after_kstep(s) runs once the MFMAs of K-step s are issued -- the operand registers of that K-step are dead then.
    auto load_frag = [&](const uint4 *wbase, int step, uint4 (&wf)[NP]) {
        const int sk = step >> 3, cb = step & 7;
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) wf[pl] = wbase[((pl * 8 + cb) * 4 + sk) << 6];
    };
    auto gemm = [&](const uint4 *wbase, const uint4 (&x)[NP][4], f32x4 (&acc)[8], auto after_kstep) {
        uint4 wf[4][NP];      // 4 sets, distance 2: a set is rewritten two steps after its last MFMA read it (no WAR wait states)
        load_frag(wbase, 0, wf[0]);
        load_frag(wbase, 1, wf[1]);
#pragma unroll
        for (int step = 0; step < 32; ++step) {
            if (step + 2 < 32) load_frag(wbase, step + 2, wf[(step + 2) & 3]);
            const int sk = step >> 3, cb = step & 7;
            f32x4 c = acc[cb];
#pragma unroll
            for (int q = 0; q < NPROD; ++q) c = Fmt<F>::mfma(wf[step & 3][Fmt<F>::PB[q]], x[Fmt<F>::PA[q]][sk], c);
            acc[cb] = c;
            if (cb == 7) after_kstep(sk);
            // issue order inside the step: the LDS reads (of the step after next) first, then this step's MFMAs
            if (step + 2 < 32) __builtin_amdgcn_sched_group_barrier(0x100, NP, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, NPROD, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // the 4 values of channel block cb -> the operand planes (K-step cb >> 1, half cb & 1)
    auto to_planes = [&](uint4 (&x)[NP][4], int cb, f32x4 v) {
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) {
            const uint32_t a = Fmt<F>::pack(v[0], v[1]), b = Fmt<F>::pack(v[2], v[3]);
            if (cb & 1) { x[pl][cb >> 1].z = a; x[pl][cb >> 1].w = b; }
            else { x[pl][cb >> 1].x = a; x[pl][cb >> 1].y = b; }
            if (pl + 1 < NP) {      // the residual x - plane, exact in fp32
                if (F == 1) {       // v_fma_mix_f32 reads the fp16 half directly: one instruction instead of convert + subtract
                    asm("v_fma_mix_f32 %0, %1, -1.0, %0 op_sel_hi:[1,0,0]" : "+v"(v[0]) : "v"(a));
                    asm("v_fma_mix_f32 %0, %1, -1.0, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(v[1]) : "v"(a));
                    asm("v_fma_mix_f32 %0, %1, -1.0, %0 op_sel_hi:[1,0,0]" : "+v"(v[2]) : "v"(b));
                    asm("v_fma_mix_f32 %0, %1, -1.0, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(v[3]) : "v"(b));
                } else {
                    const f32x2 ra = Fmt<F>::unpack(a), rb = Fmt<F>::unpack(b);
                    v[0] -= ra.x; v[1] -= ra.y; v[2] -= rb.x; v[3] -= rb.y;
                }
            }
        }
    };
    // the value of lane ^ 16 / lane ^ 32 (ds_bpermute: LDS crossbar, no memory; both lanes of an exchange form the same sum)
    const int a16 = (lane ^ 16) << 2, a32 = (lane ^ 32) << 2;
    auto xlane = [](float v, int addr) { return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(addr, __builtin_bit_cast(int, v))); };
    // GroupNorm(1, 128) of the row held by lanes r, r + 16, r + 32, r + 48 (32 channels each): two-pass, like row_gn
    // On return a[] holds the CENTRED values x - mean (what the normalisation needs next).
    auto row_stats = [&](f32x4 (&a)[8], float &rstd) {
        float s = 0.f;
#pragma unroll
        for (int cb = 0; cb < 8; ++cb) s += (a[cb][0] + a[cb][1]) + (a[cb][2] + a[cb][3]);
        s += xlane(s, a16);
        s += xlane(s, a32);
        const float mean = s * (1.0f / kC);
        float q0 = 0.f, q1 = 0.f;
#pragma unroll
        for (int cb = 0; cb < 8; ++cb) {
            a[cb] = a[cb] - f32x4{mean, mean, mean, mean};
            q0 = fmaf(a[cb][0], a[cb][0], q0); q1 = fmaf(a[cb][1], a[cb][1], q1);
            q0 = fmaf(a[cb][2], a[cb][2], q0); q1 = fmaf(a[cb][3], a[cb][3], q1);
        }
        float q = q0 + q1;
        q += xlane(q, a16);
        q += xlane(q, a32);
        rstd = 1.0f / sqrtf(q * (1.0f / kC) + p.eps);
    };

    for (; blk < nblk; blk += stride) {
        const bool live = hi_c >= 0;
        const int go = 4 * g, ro = r;
        int hi_n, wi_n;
        fetch_idx(blk + stride, hi_n, wi_n);                 // next block's indices: a block ahead
        // ---- e0 = ReLU(W_d0 d + b_d0) for this lane's 32 channels -> operand planes
        uint4 x[NP][4];
#pragma unroll
        for (int cb = 0; cb < 8; ++cb) {
            const float4 wx = l_wx[4 * cb], wy = l_wy[4 * cb], bb = l_b0[4 * cb];
            f32x4 h;
            h[0] = relu_nan(fmaf(dy, wy.x, fmaf(dx, wx.x, bb.x)));
            h[1] = relu_nan(fmaf(dy, wy.y, fmaf(dx, wx.y, bb.y)));
            h[2] = relu_nan(fmaf(dy, wy.z, fmaf(dx, wx.z, bb.z)));
            h[3] = relu_nan(fmaf(dy, wy.w, fmaf(dx, wx.w, bb.w)));
            to_planes(x, cb, h);
        }
        // ---- e1 = W_d2 e0
        f32x4 acc[8];
#pragma unroll
        for (int cb = 0; cb < 8; ++cb) acc[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
        gemm(w1, x, acc, [](int) {});
        fetch_d(hi_n, wi_n, dx, dy);                         // next block's centres (its indices have landed): under GN1 / the second GEMM
        // ---- e = ReLU(GN_d(e1)) -> operand planes (registers only); an accumulator block that has been normalised is
        // dead: U[h]'s block is requested into it (the second GEMM's accumulators start from U[h])
        const float *up = p.U + (unsigned)((live ? hi_c : 0) * kC + go);      // scalar base + 32-bit lane offset (rows < 2^24)
        {
            float rstd;
            row_stats(acc, rstd);
#pragma unroll
            for (int cb = 0; cb < 8; ++cb) {
                const float4 gg = l_gd[4 * cb], bb = l_bd[4 * cb];
                f32x4 v;
                v[0] = relu_nan(fmaf(acc[cb][0] * rstd, gg.x, bb.x));
                v[1] = relu_nan(fmaf(acc[cb][1] * rstd, gg.y, bb.y));
                v[2] = relu_nan(fmaf(acc[cb][2] * rstd, gg.z, bb.z));
                v[3] = relu_nan(fmaf(acc[cb][3] * rstd, gg.w, bb.w));
                to_planes(x, cb, v);
                acc[cb] = *reinterpret_cast<const f32x4 *>(up + 16 * cb);
            }
        }
        // ---- t = U[h] + W_c0e e + V[w]; V's blocks are requested as the K-steps release their operand registers
        f32x4 vv[8];
        const float *vp = p.V + (unsigned)((live ? wi_c : 0) * kC + go);
        gemm(w2, x, acc, [&](int sk) {
            vv[2 * sk] = *reinterpret_cast<const f32x4 *>(vp + 32 * sk);
            vv[2 * sk + 1] = *reinterpret_cast<const f32x4 *>(vp + 32 * sk + 16);
        });
#pragma unroll
        for (int cb = 0; cb < 8; ++cb) acc[cb] = acc[cb] + vv[cb];
        // ---- m = ReLU(GN_c(t))
        {
            float rstd;
            row_stats(acc, rstd);
#pragma unroll
            for (int cb = 0; cb < 8; ++cb) {
                const float4 gg = l_gc[4 * cb], bb = l_bc[4 * cb];
                acc[cb][0] = relu_nan(fmaf(acc[cb][0] * rstd, gg.x, bb.x));
                acc[cb][1] = relu_nan(fmaf(acc[cb][1] * rstd, gg.y, bb.y));
                acc[cb][2] = relu_nan(fmaf(acc[cb][2] * rstd, gg.z, bb.z));
                acc[cb][3] = relu_nan(fmaf(acc[cb][3] * rstd, gg.w, bb.w));
            }
        }
        float *mp = p.m + (int64_t)blk * (16 * kC) + (unsigned)(ro * kC + go);      // scalar base + lane offset
        if (!SEG) {
            if (live) {
#pragma unroll
                for (int cb = 0; cb < 8; ++cb) *reinterpret_cast<f32x4 *>(mp + 16 * cb) = acc[cb];
            }
        } else {
            // per-target sums inside the block: segmented suffix scan over the 16 rows (= the 16 lanes of a DPP row), keys
            // sorted; afterwards the first row of every piece holds the piece's sum.  row_shl:d = read lane + d.
            const int h1 = dpp_int_keep<0x101>(-2, hi_c), h2 = dpp_int_keep<0x102>(-2, hi_c);
            const int h4 = dpp_int_keep<0x104>(-2, hi_c), h8 = dpp_int_keep<0x108>(-2, hi_c);
            const int hp = dpp_int_keep<0x111>(-2, hi_c);                        // row_shr:1 = the previous row's key
            const float m1 = h1 == hi_c ? 1.f : 0.f, m2 = h2 == hi_c ? 1.f : 0.f, m4 = h4 == hi_c ? 1.f : 0.f, m8 = h8 == hi_c ? 1.f : 0.f;
#pragma unroll
            for (int cb = 0; cb < 8; ++cb)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float y = acc[cb][i];
                    y = fmaf(dpp_mov<0x101>(y), m1, y);
                    y = fmaf(dpp_mov<0x102>(y), m2, y);
                    y = fmaf(dpp_mov<0x104>(y), m4, y);
                    y = fmaf(dpp_mov<0x108>(y), m8, y);
                    acc[cb][i] = y;
                    if (i == 3) __builtin_amdgcn_sched_barrier(0);      // one channel block at a time (register budget)
                }
            if (live && hp != hi_c) {
#pragma unroll
                for (int cb = 0; cb < 8; ++cb) *reinterpret_cast<f32x4 *>(mp + 16 * cb) = acc[cb];
            }
        }
        hi_c = hi_n; wi_c = wi_n;
    }
}

}  // namespace lgcn

using namespace lgcn;

extern "C" int lgcn_pack_weight_kperm(const float *W, int ld, int mma, void *out, void *stream) {
    LGCN_CHECK_PTR(W); LGCN_CHECK_PTR(out);
    if (ld < kC) return LGCN_EINVAL;
    if (mma != LGCN_MMA_F16X2 && mma != LGCN_MMA_BF16) return LGCN_ESHAPE;
    LGCN_CHECK_ALIGN16(out);
    uint16_t *o = reinterpret_cast<uint16_t *>(out);
    if (mma == LGCN_MMA_F16X2) hipLaunchKernelGGL((k_pack_weight_kperm<1>), dim3(8), dim3(256), 0, (hipStream_t)stream, W, ld, o);
    else hipLaunchKernelGGL((k_pack_weight_kperm<2>), dim3(8), dim3(256), 0, (hipStream_t)stream, W, ld, o);
    return launch_status();
}

extern "C" int lgcn_att_pairs_wi(const float *agt_ctrs, const float *ctx_ctrs, const int32_t *hi, const int32_t *wi,
                                 const int32_t *n_pairs, int64_t cap, const float *wd0, const float *bd0,
                                 const float *wkd2, const float *gd, const float *btd, const float *wkc0e,
                                 const float *U, const float *V, const float *gc, const float *btc, float eps, int mma,
                                 int seg, float *m, void *stream) {
    if (mma != LGCN_MMA_F16X2 && mma != LGCN_MMA_BF16) return LGCN_ESHAPE;
    if (cap < 0 || (seg != 0 && seg != 16)) return LGCN_EINVAL;
    if (cap == 0) return LGCN_OK;
    if (cap > 0x7ffffff0) return LGCN_ESHAPE;
    const void *ptrs[] = {agt_ctrs, ctx_ctrs, hi, wi, n_pairs, wd0, bd0, wkd2, gd, btd, wkc0e, U, V, gc, btc, m};
    for (const void *q : ptrs) LGCN_CHECK_PTR(q);
    const void *al[] = {wkd2, wkc0e, U, V, m};
    for (const void *q : al) LGCN_CHECK_ALIGN16(q);
    PairParams p{agt_ctrs, ctx_ctrs, hi, wi, n_pairs, cap, wd0, bd0, wkd2, gd, btd, wkc0e, U, V, gc, btc, eps, m};
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
        cus = 256;
    const int64_t blocks = (cap + 15) / 16;       // 16-pair blocks; one wave each, 16 waves per workgroup
    const unsigned grid = (unsigned)(blocks < cus ? blocks : cus);
    hipStream_t st = (hipStream_t)stream;
    if (mma == LGCN_MMA_F16X2) {
        if (seg) hipLaunchKernelGGL((k_att_pairs_wi<1, true>), dim3(grid), dim3(1024), 0, st, p);
        else hipLaunchKernelGGL((k_att_pairs_wi<1, false>), dim3(grid), dim3(1024), 0, st, p);
    } else {
        if (seg) hipLaunchKernelGGL((k_att_pairs_wi<2, true>), dim3(grid), dim3(1024), 0, st, p);
        else hipLaunchKernelGGL((k_att_pairs_wi<2, false>), dim3(grid), dim3(1024), 0, st, p);
    }
    return launch_status();
}
