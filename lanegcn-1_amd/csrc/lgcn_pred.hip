// PredNet's tail (reference lanegcn.py:575-631 PredNet.forward, 713-737 AttDest, 147-150 the world-frame transform) in
// two launches around the row-block kernels that already run its LinearRes / Linear + GroupNorm stages:
//
//   lgcn_pred_reg    reg[a, m, t, :] = W_m h_m[a] + b_m + ctr[a]          (the M heads' nn.Linear(128, 2 T), :601-612)
//                    hd[a M + m, :]  = relu(Wd (ctr[a] - reg[a, m, T - 1, :]) + bd)     (AttDest.dist[0..1], :725-729)
//   lgcn_pred_final  cls[a, :] = sort_desc(wc . f[a M + m, :] + bc),  reg_out[a, j] = reg[a, order_j] rot[a] + orig[a]
//                    (the score nn.Linear(128, 1), the sort and the gather of :614-625, matmul + orig of Net.forward)
//
// Plain fp32 FMAs (74 MFLOP for 1,600 actors): the work is launch latency, not arithmetic -- these two launches stand
// for 6 GEMM calls, a stack, an add, a slice, a subtraction, two more GEMMs, a ReLU, a sort, an arange, an indexed gather,
// an einsum and an add of the stock path.
#include "lgcn_common.hpp"

namespace lgcn {

constexpr int kPredMaxMod = 8;
constexpr int kPredActors = 32;        // actors per workgroup of k_pred_reg
constexpr int kPredLd = 132;           // LDS row stride of the actors' feature rows (floats): 16 lanes' float4 reads spread over the banks

struct PredRegParams {
    const float *h[kPredMaxMod];       // [A, 128] per mode: the heads' LinearRes outputs
    const float *w[kPredMaxMod];       // [np2, 128] per mode
    const float *b[kPredMaxMod];       // [np2] per mode
    const float *ctrs;                 // [A, 2]
    const float *wd, *bd;              // [128, 2], [128]: AttDest.dist[0]
    float *reg;                        // [A, M, np2]
    float *hd;                         // [A * M, 128]
    int n_act, n_mod, np2;
};

__global__ __launch_bounds__(256) void k_pred_reg(const PredRegParams p) {
    __shared__ __attribute__((aligned(16))) float sW[64 * 128];
    __shared__ __attribute__((aligned(16))) float sH[kPredActors * kPredLd];
    __shared__ float sD[kPredActors][2];
    const int tid = threadIdx.x, m = blockIdx.y, a0 = blockIdx.x * kPredActors;
    const float *wm = p.w[m], *hm = p.h[m];
    for (int i = tid; i < 64 * 32; i += 256) {                  // float4 index: row i >> 5, columns 4 (i & 31)
        const int o = i >> 5;
        reinterpret_cast<float4 *>(sW)[i] = o < p.np2 ? reinterpret_cast<const float4 *>(wm)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int i = tid; i < kPredActors * 32; i += 256) {
        const int r = i >> 5, c4 = i & 31;
        const int a = a0 + r;
        *reinterpret_cast<float4 *>(sH + r * kPredLd + 4 * c4) =
            a < p.n_act ? reinterpret_cast<const float4 *>(hm + (int64_t)a * 128)[c4] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    const int al = tid & 31, og = tid >> 5;                    // actor of the block, group of 8 outputs
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    const float *hr = sH + al * kPredLd;
    for (int k = 0; k < 128; k += 4) {
        const float4 x = *reinterpret_cast<const float4 *>(hr + k);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float4 w = *reinterpret_cast<const float4 *>(sW + (og * 8 + j) * 128 + k);   // one address per half-wave
            acc[j] = fmaf(x.x, w.x, acc[j]);
            acc[j] = fmaf(x.y, w.y, acc[j]);
            acc[j] = fmaf(x.z, w.z, acc[j]);
            acc[j] = fmaf(x.w, w.w, acc[j]);
        }
    }
    const int a = a0 + al;
    float cx = 0.f, cy = 0.f;
    if (a < p.n_act) { cx = p.ctrs[2 * (int64_t)a]; cy = p.ctrs[2 * (int64_t)a + 1]; }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int o = og * 8 + j;
        if (o < p.np2) {
            const float v = (acc[j] + p.b[m][o]) + ((o & 1) ? cy : cx);
            if (a < p.n_act) p.reg[((int64_t)a * p.n_mod + m) * p.np2 + o] = v;
            if (o >= p.np2 - 2) sD[al][o & 1] = ((o & 1) ? cy : cx) - v;          // agt_ctr - dest
        }
    }
    __syncthreads();
    for (int i = tid; i < kPredActors * 32; i += 256) {
        const int r = i >> 5, c = 4 * (i & 31);
        const int ar = a0 + r;
        if (ar >= p.n_act) continue;
        const float dx = sD[r][0], dy = sD[r][1];
        float o[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float2 w = *reinterpret_cast<const float2 *>(p.wd + 2 * (c + q));
            o[q] = relu_nan(fmaf(dy, w.y, dx * w.x) + p.bd[c + q]);      // ATen's relu keeps a NaN
        }
        *reinterpret_cast<float4 *>(p.hd + ((int64_t)ar * p.n_mod + m) * 128 + c) = make_float4(o[0], o[1], o[2], o[3]);
    }
}

struct PredFinalParams {
    const float *f;                    // [A * M, 128]: the score head's LinearRes output
    const float *wc, *bc;              // [128], [1]
    const float *reg;                  // [A, M, np, 2]
    const float *rot, *orig;           // [A, 2, 2], [A, 2] or null: no transform
    float *cls;                        // [A, M] descending
    float *out;                        // [A, M, np, 2] in the order of cls
    int n_act, n_mod, np;
};

__global__ __launch_bounds__(256) void k_pred_final(const PredFinalParams p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int a = blockIdx.x * 4 + wave;                        // one wave per actor
    if (a >= p.n_act) return;
    const float w0 = p.wc[lane], w1 = p.wc[64 + lane], bc = p.bc[0];
    float c[kPredMaxMod];
#pragma unroll
    for (int m = 0; m < kPredMaxMod; ++m) {
        c[m] = 0.f;
        if (m < p.n_mod) {
            const float *fr = p.f + ((int64_t)a * p.n_mod + m) * 128;
            float s = fmaf(fr[64 + lane], w1, fr[lane] * w0);
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
            c[m] = s + bc;
        }
    }
    // descending order, equal scores in mode order, NaN scores first (torch.sort treats NaN as the largest value): a
    // total order, so every rank 0 .. n_mod - 1 is taken exactly once and every slot of cls / out is written
    int order[kPredMaxMod];
#pragma unroll
    for (int m = 0; m < kPredMaxMod; ++m) order[m] = 0;
#pragma unroll
    for (int m = 0; m < kPredMaxMod; ++m) {
        if (m < p.n_mod) {
            int rank = 0;
#pragma unroll
            for (int j = 0; j < kPredMaxMod; ++j)
                if (j < p.n_mod) {
                    const bool nj = c[j] != c[j], nm = c[m] != c[m];
                    const bool before = (nj || nm) ? (nj && (!nm || j < m)) : (c[j] > c[m] || (c[j] == c[m] && j < m));
                    if (before) ++rank;
                }
#pragma unroll
            for (int r = 0; r < kPredMaxMod; ++r)
                if (r == rank) order[r] = m;
            if (lane == 0) p.cls[(int64_t)a * p.n_mod + rank] = c[m];
        }
    }
    float r00 = 1.f, r01 = 0.f, r10 = 0.f, r11 = 1.f, ox = 0.f, oy = 0.f;
    const bool xf = p.rot != nullptr;
    if (xf) {
        const float4 r = *reinterpret_cast<const float4 *>(p.rot + 4 * (int64_t)a);
        r00 = r.x; r01 = r.y; r10 = r.z; r11 = r.w;
        ox = p.orig[2 * (int64_t)a]; oy = p.orig[2 * (int64_t)a + 1];
    }
    const int total = p.n_mod * p.np;
    for (int i = lane; i < total; i += 64) {
        const int mo = i / p.np, t = i - mo * p.np;
        int src = 0;
#pragma unroll
        for (int r = 0; r < kPredMaxMod; ++r)
            if (r == mo) src = order[r];
        const float2 v = *reinterpret_cast<const float2 *>(p.reg + (((int64_t)a * p.n_mod + src) * p.np + t) * 2);
        float2 o = v;
        if (xf) {
            o.x = fmaf(v.y, r10, v.x * r00) + ox;               // reg @ rot + orig
            o.y = fmaf(v.y, r11, v.x * r01) + oy;
        }
        *reinterpret_cast<float2 *>(p.out + (((int64_t)a * p.n_mod + mo) * p.np + t) * 2) = o;
    }
}

}  // namespace lgcn

using namespace lgcn;

extern "C" {

int lgcn_pred_reg(const lgcn_pred_reg_t *q, void *stream) {
    LGCN_CHECK_PTR(q);
    if (q->n_act < 0 || q->n_mod < 1 || q->n_mod > kPredMaxMod) return LGCN_EINVAL;
    if (q->np2 < 2 || q->np2 > 64 || (q->np2 & 1)) return LGCN_ESHAPE;
    if (q->n_act > 0x7fffffff / (kPredMaxMod * 128)) return LGCN_ESHAPE;
    PredRegParams p;
    for (int m = 0; m < kPredMaxMod; ++m) {
        p.h[m] = p.w[m] = p.b[m] = nullptr;
        if (m < q->n_mod) {
            LGCN_CHECK_PTR(q->h[m]); LGCN_CHECK_PTR(q->w[m]); LGCN_CHECK_PTR(q->b[m]);
            LGCN_CHECK_ALIGN16(q->h[m]); LGCN_CHECK_ALIGN16(q->w[m]);
            p.h[m] = q->h[m]; p.w[m] = q->w[m]; p.b[m] = q->b[m];
        }
    }
    const void *ptrs[] = {q->ctrs, q->wd, q->bd, q->reg, q->hd};
    for (const void *v : ptrs) LGCN_CHECK_PTR(v);
    LGCN_CHECK_ALIGN16(q->hd);
    if (reinterpret_cast<uintptr_t>(q->wd) & 7u) return LGCN_EALIGN;
    if (q->n_act == 0) return LGCN_OK;
    p.ctrs = q->ctrs; p.wd = q->wd; p.bd = q->bd; p.reg = q->reg; p.hd = q->hd;
    p.n_act = (int)q->n_act; p.n_mod = q->n_mod; p.np2 = q->np2;
    const unsigned gx = (unsigned)((q->n_act + kPredActors - 1) / kPredActors);
    hipLaunchKernelGGL(k_pred_reg, dim3(gx, (unsigned)q->n_mod), dim3(256), 0, (hipStream_t)stream, p);
    return launch_status();
}

int lgcn_pred_final(const float *f, const float *wc, const float *bc, const float *reg, const float *rot, const float *orig,
                    int64_t n_act, int n_mod, int n_pred, float *cls, float *out, void *stream) {
    if (n_act < 0 || n_mod < 1 || n_mod > kPredMaxMod || n_pred < 1) return LGCN_EINVAL;
    if (n_pred > 4096 || n_act > 0x7fffffff / (kPredMaxMod * 128)) return LGCN_ESHAPE;
    const void *ptrs[] = {f, wc, bc, reg, cls, out};
    for (const void *v : ptrs) LGCN_CHECK_PTR(v);
    if ((rot == nullptr) != (orig == nullptr)) return LGCN_EINVAL;
    if (reinterpret_cast<uintptr_t>(reg) & 7u || reinterpret_cast<uintptr_t>(out) & 7u) return LGCN_EALIGN;
    if (rot != nullptr) LGCN_CHECK_ALIGN16(rot);
    if (n_act == 0) return LGCN_OK;
    PredFinalParams p;
    p.f = f; p.wc = wc; p.bc = bc; p.reg = reg; p.rot = rot; p.orig = orig; p.cls = cls; p.out = out;
    p.n_act = (int)n_act; p.n_mod = n_mod; p.np = n_pred;
    hipLaunchKernelGGL(k_pred_final, dim3((unsigned)((n_act + 3) / 4)), dim3(256), 0, (hipStream_t)stream, p);
    return launch_status();
}

}  // extern "C"
