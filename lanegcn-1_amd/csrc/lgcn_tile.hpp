// Row-phase helpers shared by the f32 and the split-bf16 row-block kernels.
#pragma once
#include "lgcn_common.hpp"

namespace lgcn {

// Row phase: thread (row = t >> 3, sub = t & 7) of the 256 compute threads
// owns columns 4*sub + 32*j + {0..3}, j = 0..3 of its row (8 threads write
// 128 contiguous bytes per j when the row goes to global memory).
struct RowVals { float4 v[4]; };

__device__ __forceinline__ RowVals row_load(const float *T, int t) {
    RowVals r;
    const float *p = T + (t >> 3) * kLDA + 4 * (t & 7);
#pragma unroll
    for (int j = 0; j < 4; ++j) r.v[j] = *reinterpret_cast<const float4 *>(p + 32 * j);
    return r;
}

__device__ __forceinline__ void row_store_lds(float *T, int t, const RowVals &r) {
    float *p = T + (t >> 3) * kLDA + 4 * (t & 7);
#pragma unroll
    for (int j = 0; j < 4; ++j) *reinterpret_cast<float4 *>(p + 32 * j) = r.v[j];
}

// Cross-lane move by a DPP control word (VALU speed; __shfl_xor compiles to ds_bpermute_b32, an LDS round
// trip per step).
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, true));
}

// Sum over the 8 consecutive lanes that share a row; every lane gets the same bits as the xor-1/2/4 butterfly:
// quad_perm [1,0,3,2], quad_perm [2,3,0,1], then row_half_mirror (lane i <- lane 7 - i of its group of 8,
// which sits in the other quad and holds that quad's sum).
__device__ __forceinline__ float sum8(float x) {
    x += dpp_mov<0xB1>(x);
    x += dpp_mov<0x4E>(x);
    x += dpp_mov<0x141>(x);
    return x;
}

// GroupNorm(1, 128): per-row mean / biased variance over the 128 channels
// (layers.py:73, gcd(1, n_out) = 1 group), two-pass in registers.
__device__ __forceinline__ void row_gn(RowVals &r, int t, const float *__restrict__ g,
                                       const float *__restrict__ b, float eps) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) s += (r.v[j].x + r.v[j].y) + (r.v[j].z + r.v[j].w);
    const float mean = sum8(s) * (1.0f / kC);
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float a = r.v[j].x - mean, bb = r.v[j].y - mean, c = r.v[j].z - mean, d = r.v[j].w - mean;
        q += (a * a + bb * bb) + (c * c + d * d);
    }
    const float rstd = 1.0f / sqrtf(sum8(q) * (1.0f / kC) + eps);
    const int c0 = 4 * (t & 7);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float4 gg = *reinterpret_cast<const float4 *>(g + c0 + 32 * j);
        const float4 bb = *reinterpret_cast<const float4 *>(b + c0 + 32 * j);
        r.v[j].x = (r.v[j].x - mean) * rstd * gg.x + bb.x;
        r.v[j].y = (r.v[j].y - mean) * rstd * gg.y + bb.y;
        r.v[j].z = (r.v[j].z - mean) * rstd * gg.z + bb.z;
        r.v[j].w = (r.v[j].w - mean) * rstd * gg.w + bb.w;
    }
}

__device__ __forceinline__ void row_relu(RowVals &r) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        r.v[j].x = relu_nan(r.v[j].x); r.v[j].y = relu_nan(r.v[j].y);
        r.v[j].z = relu_nan(r.v[j].z); r.v[j].w = relu_nan(r.v[j].w);
    }
}

__device__ __forceinline__ void row_add_global(RowVals &r, const float *__restrict__ rowp, int t) {
    const float *p = rowp + 4 * (t & 7);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float4 x = *reinterpret_cast<const float4 *>(p + 32 * j);
        r.v[j].x += x.x; r.v[j].y += x.y; r.v[j].z += x.z; r.v[j].w += x.w;
    }
}

__device__ __forceinline__ void row_add(RowVals &r, const RowVals &x) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        r.v[j].x += x.v[j].x; r.v[j].y += x.v[j].y; r.v[j].z += x.v[j].z; r.v[j].w += x.v[j].w;
    }
}

__device__ __forceinline__ void row_store_global(float *__restrict__ rowp, int t, const RowVals &r) {
    float *p = rowp + 4 * (t & 7);
#pragma unroll
    for (int j = 0; j < 4; ++j) *reinterpret_cast<float4 *>(p + 32 * j) = r.v[j];
}

__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }


struct InputParams {
    const float *ctrs, *feats;
    int64_t n_rows;
    const float *wa1, *ba1, *wpa2, *ga, *bta;
    const float *ws1, *bs1, *wps2, *gs, *bts;
    float eps;
    float *out;
};

struct PairParams {
    const float *agt_ctrs, *ctx_ctrs;
    const int32_t *hi, *wi, *n_pairs;
    int64_t cap;
    const float *wd0, *bd0, *wpd2, *gd, *btd;
    const float *wpc0e, *U, *V, *gc, *btc;
    float eps;
    float *m;
};

// split-bf16 implementations (lgcn_rowmlp_bf.hip)
int agg_mlp_bf(const lgcn_agg_mlp_t &p, bool lane_conv, hipStream_t st);
int agg_mlp_pair_bf(const lgcn_agg_mlp_t &a, const lgcn_agg_mlp_t &b, hipStream_t st);
int agg_mlp_multi_bf(const lgcn_agg_mlp_t *const *ps, int n, hipStream_t st);
int mapnet_input_bf(const InputParams &p, int mma, hipStream_t st);
int att_pairs_bf(const PairParams &p, int mma, hipStream_t st);
int pack_weight_bf(const float *W, int ld, int mma, int transpose, void *out, hipStream_t st);
int pack_weight_batch_bf(const lgcn_pack_job_t *jobs, int n_jobs, int mma, hipStream_t st);

}  // namespace lgcn
