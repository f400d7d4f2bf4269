// PredLoss (reference lanegcn.py:740-807) as one forward launch and one backward launch.
//
// Per actor a (M modes, T future steps; cls [A, M], reg [A, M, T, 2], gt [A, T, 2], has [A, T] bytes):
//   last  = argmax_t( has[t] + 0.1 t / T )                 kept iff that maximum is > 1.0 (some observed step t >= 1)
//   dist_j = sqrt((reg[j, last] - gt[last])^2 summed over x, y);   min_dist, min_idx = min_j dist_j (first of equals)
//   classification (max-margin, :791-801): for every j with min_dist < cls_th and dist_j - min_dist > cls_ignore:
//       mgn_j = cls[min_idx] - cls[j];  where mgn_j < mgn:  cls_loss += cls_coef (mgn - mgn_j),  num_cls += 1
//   regression (:803-807): reg_loss += reg_coef SmoothL1(reg[min_idx, t] - gt[t]) over the steps with has[t];  num_reg += #has
// The reference runs ~60 ATen launches and two host reads for it; the comparisons that decide indices and masks are
// formed here with the same fp32 operations (no FMA: the library is built with -ffp-contract=off), so last / min_idx /
// the hinge set are the reference's.  Sums: per actor in t, j order, then a fixed-order tree over actors (no atomics:
// bitwise repeatable).  The forward keeps (min_idx, hinge bits) per actor for the backward.
#include "lgcn_common.hpp"

namespace lgcn {

constexpr int kLossMaxMod = 8, kLossMaxT = 64;

struct PredLossParams {
    const float *cls, *reg, *gt;
    const unsigned char *has;
    int64_t n_act;
    int n_mod, n_t;
    float cls_th, cls_ignore, mgn, cls_coef, reg_coef;
    float *sums;          // [2]: cls_loss, reg_loss
    int32_t *counts;      // [2]: num_cls, num_reg
    int32_t *sel;         // [A]: min_idx | hinge bits << 8, or -1 for a dropped actor
};

__global__ __launch_bounds__(1024) void k_pred_loss_fwd(const PredLossParams p) {
    __shared__ float s_f[2][1024];
    __shared__ int s_i[2][1024];
    const int tid = threadIdx.x;
    float lc = 0.f, lr = 0.f;
    int nc = 0, nr = 0;
    const int M = p.n_mod, T = p.n_t;
    for (int64_t a = tid; a < p.n_act; a += 1024) {
        const unsigned char *h = p.has + a * T;
        // last observed step: has + 0.1 t / T in fp32 as ATen forms it ((0.1f * t) / T), first maximum
        float best = -1.f;
        int last = 0;
        for (int t = 0; t < T; ++t) {
            const float v = (h[t] ? 1.f : 0.f) + (0.1f * (float)t) / (float)T;
            if (v > best) { best = v; last = t; }
        }
        int sel = -1;
        if (best > 1.0f) {
            const float gx = p.gt[(a * T + last) * 2], gy = p.gt[(a * T + last) * 2 + 1];
            float dist[kLossMaxMod];
            float dmin = 0.f;
            int jmin = 0;
            for (int j = 0; j < M; ++j) {
                const float *r = p.reg + ((a * M + j) * T + last) * 2;
                const float dx = r[0] - gx, dy = r[1] - gy;
                dist[j] = sqrtf(dx * dx + dy * dy);
                if (j == 0 || dist[j] < dmin) { dmin = dist[j]; jmin = j; }
            }
            int bits = 0;
            if (dmin < p.cls_th) {
                const float cbest = p.cls[a * M + jmin];
                for (int j = 0; j < M; ++j) {
                    if (dist[j] - dmin > p.cls_ignore) {
                        const float m = cbest - p.cls[a * M + j];
                        if (m < p.mgn) { lc += p.mgn - m; ++nc; bits |= 1 << j; }
                    }
                }
            }
            const float *r = p.reg + ((a * M + jmin) * T) * 2, *g = p.gt + a * T * 2;
            for (int t = 0; t < T; ++t) {
                if (h[t]) {
                    ++nr;
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const float d = r[2 * t + k] - g[2 * t + k], ad = fabsf(d);
                        lr += ad < 1.f ? 0.5f * d * d : ad - 0.5f;          // SmoothL1, beta = 1
                    }
                }
            }
            sel = jmin | (bits << 8);
        }
        p.sel[a] = sel;
    }
    s_f[0][tid] = lc; s_f[1][tid] = lr; s_i[0][tid] = nc; s_i[1][tid] = nr;
    __syncthreads();
    for (int w = 512; w >= 1; w >>= 1) {
        if (tid < w) {
            s_f[0][tid] += s_f[0][tid + w]; s_f[1][tid] += s_f[1][tid + w];
            s_i[0][tid] += s_i[0][tid + w]; s_i[1][tid] += s_i[1][tid + w];
        }
        __syncthreads();
    }
    if (tid == 0) {
        p.sums[0] = p.cls_coef * s_f[0][0];
        p.sums[1] = p.reg_coef * s_f[1][0];
        p.counts[0] = s_i[0][0];
        p.counts[1] = s_i[1][0];
    }
}

// dcls [A, M], dreg [A, M, T, 2]: every element is written (zero where the loss does not depend on it).
// g_cls / g_reg: the upstream gradients of cls_loss and reg_loss (device scalars).
__global__ __launch_bounds__(256) void k_pred_loss_bwd(const PredLossParams p, const float *g_cls, const float *g_reg, float *dcls,
                                                       float *dreg) {
    const int M = p.n_mod, T = p.n_t;
    const int per = M * T * 2;
    const float gc = g_cls[0] * p.cls_coef, gr = g_reg[0] * p.reg_coef;
    for (int64_t a = blockIdx.x; a < p.n_act; a += gridDim.x) {
        const int sel = p.sel[a];
        const int jmin = sel & 0xff, bits = sel < 0 ? 0 : sel >> 8;
        if ((int)threadIdx.x < M) {
            const int j = threadIdx.x;
            float d = 0.f;
            if (sel >= 0) {
                if ((bits >> j) & 1) d += gc;                       // d(mgn - (c_min - c_j)) / dc_j
                if (j == jmin) d -= gc * (float)__popc(bits);
            }
            dcls[a * M + j] = d;
        }
        for (int e = threadIdx.x; e < per; e += blockDim.x) {
            const int j = e / (2 * T), tk = e - j * 2 * T, t = tk >> 1;
            float d = 0.f;
            if (sel >= 0 && j == jmin && p.has[a * T + t]) {
                const float x = p.reg[a * per + e] - p.gt[a * T * 2 + tk];
                d = gr * (fabsf(x) < 1.f ? x : (x > 0.f ? 1.f : -1.f));
            }
            dreg[a * per + e] = d;
        }
    }
}

}  // namespace lgcn

using namespace lgcn;

static int check_loss(const float *cls, const float *reg, const float *gt, const unsigned char *has, int64_t n_act, int n_mod, int n_t,
                      const void *a, const void *b, const void *c) {
    if (n_act < 0 || n_mod < 1 || n_mod > kLossMaxMod || n_t < 1 || n_t > kLossMaxT) return LGCN_EINVAL;
    if (n_act > 0x7fffffff / (n_mod * n_t * 2)) return LGCN_ESHAPE;
    const void *ptrs[] = {cls, reg, gt, has, a, b, c};
    for (const void *q : ptrs) LGCN_CHECK_PTR(q);
    return LGCN_OK;
}

extern "C" int lgcn_pred_loss_fwd(const float *cls, const float *reg, const float *gt, const unsigned char *has, int64_t n_act,
                                  int n_mod, int n_t, float cls_th, float cls_ignore, float mgn, float cls_coef, float reg_coef,
                                  float *sums, int32_t *counts, int32_t *sel, void *stream) {
    const int rc = check_loss(cls, reg, gt, has, n_act, n_mod, n_t, sums, counts, sel);
    if (rc != LGCN_OK) return rc;
    PredLossParams p{cls, reg, gt, has, n_act, n_mod, n_t, cls_th, cls_ignore, mgn, cls_coef, reg_coef, sums, counts, sel};
    hipLaunchKernelGGL(k_pred_loss_fwd, dim3(1), dim3(1024), 0, (hipStream_t)stream, p);      // also for n_act = 0: zero sums
    return launch_status();
}

extern "C" int lgcn_pred_loss_bwd(const float *cls, const float *reg, const float *gt, const unsigned char *has, int64_t n_act,
                                  int n_mod, int n_t, float cls_coef, float reg_coef, const int32_t *sel, const float *g_cls,
                                  const float *g_reg, float *dcls, float *dreg, void *stream) {
    const int rc = check_loss(cls, reg, gt, has, n_act, n_mod, n_t, sel, g_cls, g_reg);
    if (rc != LGCN_OK) return rc;
    LGCN_CHECK_PTR(dcls); LGCN_CHECK_PTR(dreg);
    if (n_act == 0) return LGCN_OK;
    PredLossParams p{cls, reg, gt, has, n_act, n_mod, n_t, 0.f, 0.f, 0.f, cls_coef, reg_coef, nullptr, nullptr, const_cast<int32_t *>(sel)};
    const unsigned grid = (unsigned)(n_act < 4096 ? n_act : 4096);
    hipLaunchKernelGGL(k_pred_loss_bwd, dim3(grid), dim3(256), 0, (hipStream_t)stream, p, g_cls, g_reg, dcls, dreg);
    return launch_status();
}
