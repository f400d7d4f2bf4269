// Backward building blocks of the LaneGCN hot path (fp32): GroupNorm/ReLU backward rows, the
// deterministic column reductions behind dgamma/dbeta, and a row gather.  The row-GEMMs of the
// backward are lgcn_agg_mlp launches (transposed plan / transposed weights); the weight
// gradients live next to the forward gather code in lgcn_rowmlp.hip (lgcn_wgrad).
#include "lgcn_common.hpp"
#include "lgcn_tile.hpp"

namespace lgcn {

// One block = 32 rows, thread (row = t >> 3, sub = t & 7) owns channels 4*sub + 32*j + {0..3}.
__global__ __launch_bounds__(256) void k_gn_bwd(const float *__restrict__ dy, const float *__restrict__ x,
                                                const float *__restrict__ post, const float *__restrict__ gamma,
                                                int64_t n_rows, float eps, float *__restrict__ dx,
                                                float *__restrict__ dg_out, float *__restrict__ part) {
    __shared__ float red[2][32][kC + 4];
    const int t = threadIdx.x;
    const int64_t n = (int64_t)blockIdx.x * 32 + (t >> 3);
    const bool live = n < n_rows;
    const int c0 = 4 * (t & 7);
    RowVals g, xh;
#pragma unroll
    for (int j = 0; j < 4; ++j) { g.v[j] = make_float4(0.f, 0.f, 0.f, 0.f); xh.v[j] = g.v[j]; }
    if (live) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            g.v[j] = *reinterpret_cast<const float4 *>(dy + n * kC + c0 + 32 * j);
            if (post) {
                const float4 o = *reinterpret_cast<const float4 *>(post + n * kC + c0 + 32 * j);
                g.v[j].x = o.x > 0.f ? g.v[j].x : 0.f; g.v[j].y = o.y > 0.f ? g.v[j].y : 0.f;
                g.v[j].z = o.z > 0.f ? g.v[j].z : 0.f; g.v[j].w = o.w > 0.f ? g.v[j].w : 0.f;
            }
            if (dg_out) *reinterpret_cast<float4 *>(dg_out + n * kC + c0 + 32 * j) = g.v[j];
        }
    }
    if (gamma == nullptr) {   // plain mask
        if (live)
#pragma unroll
            for (int j = 0; j < 4; ++j) *reinterpret_cast<float4 *>(dx + n * kC + c0 + 32 * j) = g.v[j];
        return;
    }
    float rstd = 0.f;
    if (live) {
#pragma unroll
        for (int j = 0; j < 4; ++j) xh.v[j] = *reinterpret_cast<const float4 *>(x + n * kC + c0 + 32 * j);
    }
    {   // xhat = (x - mean) * rstd, two-pass like the forward
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) s += (xh.v[j].x + xh.v[j].y) + (xh.v[j].z + xh.v[j].w);
        const float mean = sum8(s) * (1.0f / kC);
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            xh.v[j].x -= mean; xh.v[j].y -= mean; xh.v[j].z -= mean; xh.v[j].w -= mean;
            q += (xh.v[j].x * xh.v[j].x + xh.v[j].y * xh.v[j].y) + (xh.v[j].z * xh.v[j].z + xh.v[j].w * xh.v[j].w);
        }
        rstd = 1.0f / sqrtf(sum8(q) * (1.0f / kC) + eps);
#pragma unroll
        for (int j = 0; j < 4; ++j) { xh.v[j].x *= rstd; xh.v[j].y *= rstd; xh.v[j].z *= rstd; xh.v[j].w *= rstd; }
    }
    float m1 = 0.f, m2 = 0.f;
    RowVals d;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float4 gm = *reinterpret_cast<const float4 *>(gamma + c0 + 32 * j);
        d.v[j] = make_float4(g.v[j].x * gm.x, g.v[j].y * gm.y, g.v[j].z * gm.z, g.v[j].w * gm.w);
        m1 += (d.v[j].x + d.v[j].y) + (d.v[j].z + d.v[j].w);
        m2 += (d.v[j].x * xh.v[j].x + d.v[j].y * xh.v[j].y) + (d.v[j].z * xh.v[j].z + d.v[j].w * xh.v[j].w);
    }
    m1 = sum8(m1) * (1.0f / kC);
    m2 = sum8(m2) * (1.0f / kC);
    if (live) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float4 o;
            o.x = rstd * (d.v[j].x - m1 - xh.v[j].x * m2); o.y = rstd * (d.v[j].y - m1 - xh.v[j].y * m2);
            o.z = rstd * (d.v[j].z - m1 - xh.v[j].z * m2); o.w = rstd * (d.v[j].w - m1 - xh.v[j].w * m2);
            *reinterpret_cast<float4 *>(dx + n * kC + c0 + 32 * j) = o;
        }
    }
    // per-block partial dgamma / dbeta: sum over the block's 32 rows (fixed order)
    const int row = t >> 3;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float *a = &red[0][row][c0 + 32 * j], *b = &red[1][row][c0 + 32 * j];
        a[0] = g.v[j].x * xh.v[j].x; a[1] = g.v[j].y * xh.v[j].y; a[2] = g.v[j].z * xh.v[j].z; a[3] = g.v[j].w * xh.v[j].w;
        b[0] = g.v[j].x; b[1] = g.v[j].y; b[2] = g.v[j].z; b[3] = g.v[j].w;
    }
    __syncthreads();
    {
        const int which = t >> 7, c = t & 127;
        float s = 0.f;
#pragma unroll 8
        for (int rr = 0; rr < 32; ++rr) s += red[which][rr][c];
        part[((int64_t)which * gridDim.x + blockIdx.x) * kC + c] = s;
    }
}

// out[which][c] = sum_b part[which][b][c]   (2 x 128 outputs, one block of 256 threads, fixed order)
__global__ __launch_bounds__(256) void k_colsum_parts(const float *__restrict__ part, int n_blocks,
                                                      float *__restrict__ dgamma, float *__restrict__ dbeta) {
    const int which = threadIdx.x >> 7, c = threadIdx.x & 127;
    const float *p = part + (int64_t)which * n_blocks * kC + c;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int b = 0;
    for (; b + 3 < n_blocks; b += 4) {
        s0 += p[(int64_t)b * kC]; s1 += p[(int64_t)(b + 1) * kC]; s2 += p[(int64_t)(b + 2) * kC]; s3 += p[(int64_t)(b + 3) * kC];
    }
    for (; b < n_blocks; ++b) s0 += p[(int64_t)b * kC];
    float *o = which == 0 ? dgamma : dbeta;
    if (o) o[c] = (s0 + s1) + (s2 + s3);
}

__global__ __launch_bounds__(256) void k_gather_rows(const float4 *__restrict__ src, const int32_t *__restrict__ idx,
                                                     const int32_t *__restrict__ n_dev, int64_t cap,
                                                     float4 *__restrict__ out) {
    int64_t n = *n_dev;
    if (n < 0 || n > cap) n = cap;
    const int l = threadIdx.x & 31;
    for (int64_t i = (int64_t)blockIdx.x * 8 + (threadIdx.x >> 5); i < n; i += (int64_t)gridDim.x * 8)
        out[i * 32 + l] = src[(int64_t)idx[i] * 32 + l];
}

__global__ __launch_bounds__(256) void k_gn_fwd(const float *__restrict__ x, const float *__restrict__ gamma,
                                                const float *__restrict__ beta, const float *__restrict__ res,
                                                int64_t n_rows, float eps, int relu, float *__restrict__ out) {
    const int t = threadIdx.x;
    const int64_t n = (int64_t)blockIdx.x * 32 + (t >> 3);
    const bool live = n < n_rows;
    const int c0 = 4 * (t & 7);
    RowVals r;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        r.v[j] = live ? *reinterpret_cast<const float4 *>(x + n * kC + c0 + 32 * j) : make_float4(0.f, 0.f, 0.f, 0.f);
    if (gamma) row_gn(r, t, gamma, beta, eps);
    if (live && res) row_add_global(r, res + n * kC, t);
    if (relu) row_relu(r);
    if (live) row_store_global(out + n * kC, t, r);
}

// One wave per item of x [n, C, L]: GroupNorm over the item's C * L elements (two-pass mean / biased variance),
// per-channel affine, optional residual and ReLU.  The item is a few KB (640 .. 2,560 floats in ActorNet) and is
// re-read from cache for the second and third sweep.
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

__global__ __launch_bounds__(256) void k_gn_cl(const float *__restrict__ x, int64_t n_items, int C, int L,
                                               const float *__restrict__ gamma, const float *__restrict__ beta, float eps,
                                               const float *__restrict__ res, int res_up2, int relu, int cl,
                                               float *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t item = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= n_items) return;
    const int n = C * L;
    const float *xi = x + item * n;
    float s = 0.f;
    for (int i = lane; i < n; i += 64) s += xi[i];
    const float mean = wave_sum(s) / (float)n;
    float q = 0.f;
    for (int i = lane; i < n; i += 64) { const float d = xi[i] - mean; q += d * d; }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)n + eps);
    const int Lh = L >> 1;
    const float *ri = res ? res + item * (res_up2 ? C * Lh : n) : nullptr;
    float *oi = out + item * n;
    for (int i = lane; i < n; i += 64) {
        const int c = cl ? i % C : i / L;
        float v = (xi[i] - mean) * rstd * gamma[c] + beta[c];
        if (ri && res_up2) {
            const int t = cl ? i / C : i - c * L, h = t >> 1;
            const int hn = (t & 1) ? (h + 1 < Lh ? h + 1 : Lh - 1) : (h > 0 ? h - 1 : 0);     // the neighbour that contributes 1/4
            const float mid = cl ? ri[h * C + c] : ri[c * Lh + h];
            const float nb = cl ? ri[hn * C + c] : ri[c * Lh + hn];
            const float up = (t & 1) ? 0.75f * mid + 0.25f * nb : 0.25f * nb + 0.75f * mid;
            v = up + v;
        } else if (ri) v += ri[i];
        if (relu) v = relu_nan(v);
        oi[i] = v;
    }
}

// Fast path of k_gn_cl for items of at most 64 * 4 * NV floats whose 4-element chunks stay inside one channel run
// (channels_last with C % 4 == 0: four consecutive channels; NCL with L % 4 == 0: four consecutive positions of one
// channel): the item is read ONCE as float4 into registers, normalised and written back -- the generic kernel sweeps
// it three times with an integer division per element (15 us per call at [1600, 128, 20] against ~4 here).
template <int NV>
__global__ __launch_bounds__(256) void k_gn_cl_vec(const float *__restrict__ x, int64_t n_items, int C, int L,
                                                   const float *__restrict__ gamma, const float *__restrict__ beta,
                                                   float eps, const float *__restrict__ res, int res_up2, int relu, int cl,
                                                   float *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t item = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= n_items) return;
    const int n = C * L, nq = n >> 2;                  // float4 chunks per item
    const float4 *xi = reinterpret_cast<const float4 *>(x + item * n);
    float4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int j = lane + 64 * k;
        v[k] = j < nq ? xi[j] : make_float4(0.f, 0.f, 0.f, 0.f);
        s += (v[k].x + v[k].y) + (v[k].z + v[k].w);
    }
    const float mean = wave_sum(s) / (float)n;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        if (lane + 64 * k < nq) {
            const float a = v[k].x - mean, b = v[k].y - mean, c = v[k].z - mean, d = v[k].w - mean;
            q += (a * a + b * b) + (c * c + d * d);
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)n + eps);
    const int Lh = L >> 1;
    const float *ri = res ? res + item * (res_up2 ? C * Lh : n) : nullptr;
    float4 *oi = reinterpret_cast<float4 *>(out + item * n);
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int j = lane + 64 * k;
        if (j >= nq) continue;
        const int e = 4 * j;
        float4 g4, b4, r4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (cl) {                                       // 4 consecutive channels at position t
            const int t = e / C, c = e - t * C;
            g4 = *reinterpret_cast<const float4 *>(gamma + c);
            b4 = *reinterpret_cast<const float4 *>(beta + c);
            if (ri && res_up2) {
                const int h = t >> 1, hn = (t & 1) ? (h + 1 < Lh ? h + 1 : Lh - 1) : (h > 0 ? h - 1 : 0);
                const float4 mid = *reinterpret_cast<const float4 *>(ri + h * C + c);
                const float4 nb = *reinterpret_cast<const float4 *>(ri + hn * C + c);
                if (t & 1) r4 = make_float4(0.75f * mid.x + 0.25f * nb.x, 0.75f * mid.y + 0.25f * nb.y,
                                            0.75f * mid.z + 0.25f * nb.z, 0.75f * mid.w + 0.25f * nb.w);
                else r4 = make_float4(0.25f * nb.x + 0.75f * mid.x, 0.25f * nb.y + 0.75f * mid.y,
                                      0.25f * nb.z + 0.75f * mid.z, 0.25f * nb.w + 0.75f * mid.w);
            } else if (ri) r4 = *reinterpret_cast<const float4 *>(ri + e);
        } else {                                        // 4 consecutive positions of channel c (no upsampling here)
            const int c = e / L;
            const float gc = gamma[c], bc = beta[c];
            g4 = make_float4(gc, gc, gc, gc);
            b4 = make_float4(bc, bc, bc, bc);
            if (ri) r4 = *reinterpret_cast<const float4 *>(ri + e);
        }
        float4 y;
        y.x = (v[k].x - mean) * rstd * g4.x + b4.x; y.y = (v[k].y - mean) * rstd * g4.y + b4.y;
        y.z = (v[k].z - mean) * rstd * g4.z + b4.z; y.w = (v[k].w - mean) * rstd * g4.w + b4.w;
        if (ri && res_up2) { y.x = r4.x + y.x; y.y = r4.y + y.y; y.z = r4.z + y.z; y.w = r4.w + y.w; }
        else if (ri) { y.x += r4.x; y.y += r4.y; y.z += r4.z; y.w += r4.w; }
        if (relu) { y.x = relu_nan(y.x); y.y = relu_nan(y.y); y.z = relu_nan(y.z); y.w = relu_nan(y.w); }
        oi[j] = y;
    }
}

// Backward of k_gn_cl, one wave per item: statistics recomputed, two item-wide sums (DPP-free wave_sum), then dx
// element-wise and the per-channel sums of this item with one lane per channel (fixed order: deterministic).
__global__ __launch_bounds__(256) void k_gn_cl_bwd(const float *__restrict__ dy, const float *__restrict__ x,
                                                   const float *__restrict__ post, const float *__restrict__ gamma,
                                                   int64_t n_items, int C, int L, float eps, float *__restrict__ dx,
                                                   float *__restrict__ g_out, float *__restrict__ part) {
    const int lane = threadIdx.x & 63;
    const int64_t item = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= n_items) return;
    const int n = C * L;
    const float *xi = x + item * n, *di = dy + item * n;
    const float *pi = post ? post + item * n : nullptr;
    float s = 0.f;
    for (int i = lane; i < n; i += 64) s += xi[i];
    const float mean = wave_sum(s) / (float)n;
    float q = 0.f;
    for (int i = lane; i < n; i += 64) { const float d = xi[i] - mean; q += d * d; }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)n + eps);
    float s1 = 0.f, s2 = 0.f;
    for (int i = lane; i < n; i += 64) {
        const float gv = (pi == nullptr || pi[i] > 0.f) ? di[i] : 0.f;
        const float gg = gv * gamma[i / L];
        s1 += gg;
        s2 += gg * ((xi[i] - mean) * rstd);
    }
    const float m1 = wave_sum(s1) / (float)n, m2 = wave_sum(s2) / (float)n;
    float *dxi = dx + item * n;
    float *gi = g_out ? g_out + item * n : nullptr;
    for (int i = lane; i < n; i += 64) {
        const float gv = (pi == nullptr || pi[i] > 0.f) ? di[i] : 0.f;
        const float xh = (xi[i] - mean) * rstd;
        dxi[i] = rstd * (gv * gamma[i / L] - m1 - xh * m2);
        if (gi) gi[i] = gv;
    }
    float *pt = part + item * 2 * C;
    for (int c = lane; c < C; c += 64) {
        float a = 0.f, b = 0.f;
        for (int l = 0; l < L; ++l) {
            const int i = c * L + l;
            const float gv = (pi == nullptr || pi[i] > 0.f) ? di[i] : 0.f;
            a += gv * ((xi[i] - mean) * rstd);
            b += gv;
        }
        pt[c] = a;
        pt[C + c] = b;
    }
}

// one half-wave per output row, float4 per lane
__global__ __launch_bounds__(256) void k_gather_sum(const float4 *__restrict__ src, const int32_t *__restrict__ rowptr,
                                                    const int32_t *__restrict__ col, int64_t n_rows,
                                                    float4 *__restrict__ out) {
    const int l = threadIdx.x & 31;
    for (int64_t n = (int64_t)blockIdx.x * 8 + (threadIdx.x >> 5); n < n_rows; n += (int64_t)gridDim.x * 8) {
        const int b = rowptr[n], e = rowptr[n + 1];
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        int j = b;
        for (; j + 3 < e; j += 4) {
            float4 v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = src[(int64_t)(col ? col[j + q] : j + q) * 32 + l];
            s = f4add(f4add(f4add(f4add(s, v[0]), v[1]), v[2]), v[3]);
        }
        for (; j < e; ++j) s = f4add(s, src[(int64_t)(col ? col[j] : j) * 32 + l]);
        out[n * 32 + l] = s;
    }
}

__global__ __launch_bounds__(256) void k_pair_add(const float4 *__restrict__ c, const float4 *__restrict__ U,
                                                  const int32_t *__restrict__ hi, const float4 *__restrict__ V,
                                                  const int32_t *__restrict__ wi, const int32_t *__restrict__ n_dev,
                                                  int64_t cap, float4 *__restrict__ out) {
    int64_t n = *n_dev;
    if (n < 0 || n > cap) n = cap;
    const int l = threadIdx.x & 31;
    for (int64_t i = (int64_t)blockIdx.x * 8 + (threadIdx.x >> 5); i < n; i += (int64_t)gridDim.x * 8)
        out[i * 32 + l] = f4add(f4add(c[i * 32 + l], U[(int64_t)hi[i] * 32 + l]), V[(int64_t)wi[i] * 32 + l]);
}

// Any element of x not finite -> flag[0] |= bit (integer atomic, only when something is found).  Up to two
// tensors per launch.  The 16-bit-plane matrix modes have fp16's / bf16's range: an overflow shows up as NaN rows
// in the stage outputs (the kernels' ReLU keeps NaN), this is how a caller looks for them without a device->host
// copy of the features.
__global__ __launch_bounds__(256) void k_check_finite(const float *__restrict__ a, int64_t na, const float *__restrict__ b,
                                                      int64_t nb, int32_t *flag, int bit) {
    bool bad = false;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x, i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    // finite <=> (v - v) == 0 (inf - inf and NaN - NaN are NaN); whole float4s first, then the <= 3 leftover floats
    const int64_t na4 = na >> 2, nb4 = nb >> 2;
    for (int64_t i = i0; i < na4 + nb4; i += stride) {
        const float4 v = i < na4 ? reinterpret_cast<const float4 *>(a)[i] : reinterpret_cast<const float4 *>(b)[i - na4];
        const float t = (v.x - v.x) + (v.y - v.y) + (v.z - v.z) + (v.w - v.w);
        bad |= !(t == 0.f);
    }
    const int64_t ra = na - 4 * na4, rb = nb - 4 * nb4;
    if (i0 < ra + rb) {
        const float v = i0 < ra ? a[4 * na4 + i0] : b[4 * nb4 + (i0 - ra)];
        bad |= !((v - v) == 0.f);
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, bit);
}

}  // namespace lgcn

using namespace lgcn;

extern "C" {

int lgcn_gn_bwd(const float *dy, const float *x, const float *post, const float *gamma, int64_t n_rows, float eps,
                float *dx, float *dg_out, float *dgamma, float *dbeta, float *part, void *stream) {
    if (n_rows < 0) return LGCN_EINVAL;
    if (n_rows == 0) return LGCN_OK;
    if (n_rows > 0x7fffffff) return LGCN_ESHAPE;
    LGCN_CHECK_PTR(dy); LGCN_CHECK_PTR(dx);
    LGCN_CHECK_ALIGN16(dy); LGCN_CHECK_ALIGN16(dx);
    if (post) LGCN_CHECK_ALIGN16(post);
    if (dg_out) LGCN_CHECK_ALIGN16(dg_out);
    if (gamma) {
        LGCN_CHECK_PTR(x); LGCN_CHECK_PTR(part);
        LGCN_CHECK_ALIGN16(x); LGCN_CHECK_ALIGN16(gamma);
    }
    const int nb = (int)((n_rows + 31) / 32);
    hipLaunchKernelGGL(k_gn_bwd, dim3(nb), dim3(256), 0, (hipStream_t)stream, dy, x, post, gamma, n_rows, eps, dx, dg_out, part);
    if (gamma && (dgamma || dbeta))
        hipLaunchKernelGGL(k_colsum_parts, dim3(1), dim3(256), 0, (hipStream_t)stream, part, nb, dgamma, dbeta);
    return launch_status();
}

int lgcn_gn_fwd(const float *x, const float *gamma, const float *beta, const float *res, int64_t n_rows, float eps,
                int relu, float *out, void *stream) {
    if (n_rows < 0) return LGCN_EINVAL;
    if (n_rows == 0) return LGCN_OK;
    if (n_rows > 0x7fffffff) return LGCN_ESHAPE;
    LGCN_CHECK_PTR(x); LGCN_CHECK_PTR(out);
    LGCN_CHECK_ALIGN16(x); LGCN_CHECK_ALIGN16(out);
    if (gamma) { LGCN_CHECK_PTR(beta); LGCN_CHECK_ALIGN16(gamma); LGCN_CHECK_ALIGN16(beta); }
    if (res) LGCN_CHECK_ALIGN16(res);
    hipLaunchKernelGGL(k_gn_fwd, dim3((unsigned)((n_rows + 31) / 32)), dim3(256), 0, (hipStream_t)stream, x, gamma, beta,
                       res, n_rows, eps, relu, out);
    return launch_status();
}

int lgcn_gn_cl(const float *x, int64_t n_items, int C, int L, const float *gamma, const float *beta, float eps,
               const float *res, int res_up2, int relu, int channels_last, float *out, void *stream) {
    if (n_items < 0 || C < 1 || L < 1 || (int64_t)C * L > 16384) return LGCN_EINVAL;
    if (res_up2 && (res == nullptr || (L & 1))) return LGCN_EINVAL;
    if (n_items == 0) return LGCN_OK;
    if (n_items > 0x7fffffff) return LGCN_ESHAPE;
    LGCN_CHECK_PTR(x); LGCN_CHECK_PTR(gamma); LGCN_CHECK_PTR(beta); LGCN_CHECK_PTR(out);
    const int n = C * L;
    const bool aligned = (((uintptr_t)x | (uintptr_t)out | (uintptr_t)(res ? res : x)) & 15) == 0 &&
                         (channels_last ? ((uintptr_t)gamma | (uintptr_t)beta) & 15 : 0) == 0;
    const bool vec = aligned && n <= 64 * 4 * 16 &&
                     (channels_last ? (C % 4 == 0) : (L % 4 == 0 && !res_up2));
    const dim3 grid((unsigned)((n_items + 3) / 4));
    if (vec && n <= 64 * 4 * 10)
        hipLaunchKernelGGL((k_gn_cl_vec<10>), grid, dim3(256), 0, (hipStream_t)stream, x, n_items, C, L, gamma, beta, eps,
                           res, res_up2, relu, channels_last, out);
    else if (vec)
        hipLaunchKernelGGL((k_gn_cl_vec<16>), grid, dim3(256), 0, (hipStream_t)stream, x, n_items, C, L, gamma, beta, eps,
                           res, res_up2, relu, channels_last, out);
    else
        hipLaunchKernelGGL(k_gn_cl, grid, dim3(256), 0, (hipStream_t)stream, x, n_items, C, L, gamma, beta, eps, res,
                           res_up2, relu, channels_last, out);
    return launch_status();
}

int lgcn_gn_cl_bwd(const float *dy, const float *x, const float *post, const float *gamma, int64_t n_items, int C,
                   int L, float eps, float *dx, float *g, float *part, void *stream) {
    if (n_items < 0 || C < 1 || L < 1 || (int64_t)C * L > 16384) return LGCN_EINVAL;
    if (n_items == 0) return LGCN_OK;
    if (n_items > 0x7fffffff) return LGCN_ESHAPE;
    LGCN_CHECK_PTR(dy); LGCN_CHECK_PTR(x); LGCN_CHECK_PTR(gamma); LGCN_CHECK_PTR(dx); LGCN_CHECK_PTR(part);
    hipLaunchKernelGGL(k_gn_cl_bwd, dim3((unsigned)((n_items + 3) / 4)), dim3(256), 0, (hipStream_t)stream, dy, x, post,
                       gamma, n_items, C, L, eps, dx, g, part);
    return launch_status();
}

int lgcn_gather_sum(const float *src, const int32_t *rowptr, const int32_t *col, int64_t n_rows, float *out,
                    void *stream) {
    if (n_rows < 0) return LGCN_EINVAL;
    if (n_rows == 0) return LGCN_OK;
    LGCN_CHECK_PTR(src); LGCN_CHECK_PTR(rowptr); LGCN_CHECK_PTR(out);
    LGCN_CHECK_ALIGN16(src); LGCN_CHECK_ALIGN16(out);
    int64_t blocks = (n_rows + 7) / 8;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_gather_sum, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4 *>(src), rowptr, col, n_rows, reinterpret_cast<float4 *>(out));
    return launch_status();
}

int lgcn_pair_add(const float *c, const float *U, const int32_t *hi, const float *V, const int32_t *wi,
                  const int32_t *n_dev, int64_t cap, float *out, void *stream) {
    if (cap < 0) return LGCN_EINVAL;
    if (cap == 0) return LGCN_OK;
    const void *ptrs[] = {c, U, hi, V, wi, n_dev, out};
    for (const void *q : ptrs) LGCN_CHECK_PTR(q);
    const void *al[] = {c, U, V, out};
    for (const void *q : al) LGCN_CHECK_ALIGN16(q);
    int64_t blocks = (cap + 7) / 8;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_pair_add, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4 *>(c), reinterpret_cast<const float4 *>(U), hi,
                       reinterpret_cast<const float4 *>(V), wi, n_dev, cap, reinterpret_cast<float4 *>(out));
    return launch_status();
}

int lgcn_gather_rows(const float *src, const int32_t *idx, const int32_t *n_dev, int64_t cap, float *out, void *stream) {
    if (cap < 0) return LGCN_EINVAL;
    if (cap == 0) return LGCN_OK;
    LGCN_CHECK_PTR(src); LGCN_CHECK_PTR(idx); LGCN_CHECK_PTR(n_dev); LGCN_CHECK_PTR(out);
    LGCN_CHECK_ALIGN16(src); LGCN_CHECK_ALIGN16(out);
    int64_t blocks = (cap + 7) / 8;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_gather_rows, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4 *>(src), idx, n_dev, cap, reinterpret_cast<float4 *>(out));
    return launch_status();
}

int lgcn_check_finite(const float *a, int64_t na, const float *b, int64_t nb, int32_t *flag, int bit, void *stream) {
    if (na < 0 || nb < 0 || bit == 0) return LGCN_EINVAL;
    if (na + nb == 0) return LGCN_OK;
    LGCN_CHECK_PTR(flag);
    if (na) { LGCN_CHECK_PTR(a); LGCN_CHECK_ALIGN16(a); }
    if (nb) { LGCN_CHECK_PTR(b); LGCN_CHECK_ALIGN16(b); }
    int64_t blocks = ((na + nb) / 4 + 1023) / 1024;      // four float4 per thread
    if (blocks < 1) blocks = 1;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_check_finite, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, na, b, nb, flag, bit);
    return launch_status();
}

}  // extern "C"
