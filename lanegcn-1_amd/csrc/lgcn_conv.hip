// ActorNet's building block (reference layers.py:40-62 Conv1d, 142-190 Res1d; lanegcn.py:212-263) in ONE launch:
//
//   out[a, l, :] = act( GN_{(C, L) of actor a}( sum_t W_t x[a, l * stride + t - pad, :] ) + residual )
//
// on channels-last tensors x [A, Lin, Cin], out [A, Lout, Cout] (Lout = (Lin + 2 pad - K) / stride + 1, pad = (K - 1) / 2,
// K = 1 or 3, stride = 1 or 2, Cin <= 128, Cout in {32, 64, 128}, GroupNorm with one group = statistics over all
// Lout x Cout values of an actor).  residual: none | a tensor of the output's shape | a tensor of half the length,
// upsampled x2 on the fly (F.interpolate(mode="linear", align_corners=False): the FPN's top-down step).
//
// A workgroup owns NA whole actors = 80 output rows (NA = 80 / Lout: 4, 8 or 16 actors at Lout = 20, 10, 5), so the
// GroupNorm statistics never leave the CU.  The actors' input rows are staged once in LDS as fp16 operand planes
// (2 planes, 3 products: the fp32-grade split of lgcn_mma_bf.hpp); the convolution is K x ceil(Cin / 32) MFMA K-steps
// whose A fragments are the staged rows shifted by the tap (rows outside the sequence read an all-zero row); a wave
// owns one 16-channel block of the output and a share of the five 16-row sub-blocks, its weight fragments come from
// the packed image (lgcn_conv_pack_weight) once per K-step.  The 80 x Cout fp32 tile then goes through LDS to the
// norm: 512 / NA threads per actor, two passes (mean, then variance about it, as ATen's GroupNorm), residual, ReLU, and
// 512-byte coalesced stores.
#include "lgcn_common.hpp"
#include "lgcn_tile.hpp"
#include "lgcn_mma_bf.hpp"

namespace lgcn {

constexpr int kConvRows = 80;          // output rows per workgroup (5 sub-blocks of 16)
constexpr int kConvSub = kConvRows / 16;

struct ConvParams {
    const float *x;                    // [A, lin, cin]
    int64_t n_act;
    int lin, cin, cout, ks, stride, lout;
    const uint4 *wp;                   // packed weight image
    const float *gamma, *beta;
    float eps;
    const float *res;                  // residual source or null
    int res_mode;                      // 0 none, 1 [A, lout, cout], 2 [A, lout / 2, cout] upsampled x2
    int relu;
    float *out;                        // [A, lout, cout]
};

__host__ __device__ inline int conv_kpad(int cin) { return (cin + 31) & ~31; }

// Packed image: for tap t, K-chunk kc (32 input channels), channel block cb (16 outputs), plane pl: 64 x uint4, lane
// (n = lane & 15, kq = lane >> 4) holds W[16 cb + n][32 kc + 8 kq + j][t], j = 0..7, as fp16 plane pl (hi, then the
// rounding of the residual).  Input channels beyond cin are zero.
__global__ __launch_bounds__(256) void k_conv_pack(const float *w, int cout, int cin, int ks, uint4 *out) {
    const int nkc = conv_kpad(cin) / 32, ncb = cout / 16;
    const int64_t total = (int64_t)ks * nkc * ncb * 64;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int lane = (int)(i & 63);
    int64_t q = i >> 6;
    const int cb = (int)(q % ncb); q /= ncb;
    const int kc = (int)(q % nkc);
    const int t = (int)(q / nkc);
    const int n = lane & 15, kq = lane >> 4;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = 32 * kc + 8 * kq + j;
        v[j] = c < cin ? w[((int64_t)(16 * cb + n) * cin + c) * ks + t] : 0.f;
    }
    uint32_t hi[4], lo[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        hi[j] = Fmt<1>::pack(v[2 * j], v[2 * j + 1]);
        const f32x2 r = Fmt<1>::unpack(hi[j]);
        lo[j] = Fmt<1>::pack(v[2 * j] - r.x, v[2 * j + 1] - r.y);
    }
    const int64_t base = ((((int64_t)t * nkc + kc) * ncb + cb) * 2) << 6;
    out[base + lane] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
    out[base + 64 + lane] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
}

// F.interpolate(scale_factor = 2, mode = "linear", align_corners = False) of a length-n sequence at output position j:
// source coordinate (j + 0.5) / 2 - 0.5, clamped at 0; weights 0.75 / 0.25 (and 1 / 0 at the two ends).
__device__ __forceinline__ void up2_taps(int j, int n, int &i0, int &i1, float &w1) {
    float src = (j + 0.5f) * 0.5f - 0.5f;
    src = src < 0.f ? 0.f : src;
    i0 = (int)src;
    i1 = i0 + 1 < n ? i0 + 1 : n - 1;
    w1 = src - (float)i0;
}

template <int KS, int NKC>                                     // taps, 32-channel K chunks; KS == 0: both read from p (any shape)
__global__ __launch_bounds__(512) void k_conv_gn(const ConvParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int na = kConvRows / p.lout;                        // actors per workgroup
    const int64_t a0 = (int64_t)blockIdx.x * na;
    const int kpad = conv_kpad(p.cin), ldk = kpad + 8;         // fp16 elements per staged row (+ 16 B: bank spread)
    const int n_in = na * p.lin;                               // staged input rows; row n_in is all zero
    uint16_t *P0 = reinterpret_cast<uint16_t *>(smem);
    uint16_t *P1 = P0 + (n_in + 1) * ldk;
    const int ldt = p.cout + 4;
    float *T = reinterpret_cast<float *>(smem);                // the fp32 tile takes the planes' place once the GEMM is done
    const int pad = (p.ks - 1) >> 1;

    // wave -> channel block cb and the row sub-blocks rb0, rb0 + nw, ...
    const int ncb = p.cout >> 4, nw = 8 / ncb, nkc = KS ? NKC : kpad >> 5;
    const int cb = wave % ncb, rb0 = wave / ncb;
    const int kq = lane >> 4;
    auto wfrag = [&](int s_, uint4 &h, uint4 &l_) {               // packed weight fragments of K-step s_ = t * nkc + kc
        const int64_t wb = (((int64_t)s_ * ncb + cb) * 2) << 6;
        h = p.wp[wb + lane];
        l_ = p.wp[wb + 64 + lane];
    };
    // An L2 round trip is several K-steps long (a K-step is <= 15 MFMAs): with the shape known the first kWd steps'
    // fragments are requested before the rows are staged and the ring is refilled kWd steps ahead.
    constexpr int NKS = KS * NKC, kWd = NKS < 6 ? (NKS ? NKS : 1) : 6;
    uint4 wh[kWd], wl[kWd];
    if constexpr (KS != 0) {
#pragma unroll
        for (int s_ = 0; s_ < kWd; ++s_) wfrag(s_, wh[s_], wl[s_]);
    }

    // ---- stage the actors' input rows as two fp16 planes (4 channels per thread and step, four row loads in flight)
    {
        const int c4n = kpad / 4, total = (n_in + 1) * c4n;
        for (int i0 = tid; i0 < total; i0 += 4 * 512) {
            float4 v[4];
            int rr[4], cc[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * 512;
                const int r = i / c4n, c = 4 * (i - r * c4n);
                rr[u] = r; cc[u] = c;
                v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                const int ar = r / p.lin;
                const int64_t a = a0 + ar;
                if (i < total && r < n_in && a < p.n_act) {
                    const float *src = p.x + (a * p.lin + (r - ar * p.lin)) * p.cin + c;
                    if (c + 3 < p.cin && (p.cin & 3) == 0) v[u] = *reinterpret_cast<const float4 *>(src);
                    else {
                        if (c < p.cin) v[u].x = src[0];
                        if (c + 1 < p.cin) v[u].y = src[1];
                        if (c + 2 < p.cin) v[u].z = src[2];
                        if (c + 3 < p.cin) v[u].w = src[3];
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (i0 + u * 512 < total) {
                    const uint32_t h0 = Fmt<1>::pack(v[u].x, v[u].y), h1 = Fmt<1>::pack(v[u].z, v[u].w);
                    const f32x2 r0 = Fmt<1>::unpack(h0), r1 = Fmt<1>::unpack(h1);
                    *reinterpret_cast<uint2 *>(P0 + rr[u] * ldk + cc[u]) = make_uint2(h0, h1);
                    *reinterpret_cast<uint2 *>(P1 + rr[u] * ldk + cc[u]) =
                        make_uint2(Fmt<1>::pack(v[u].x - r0.x, v[u].y - r0.y), Fmt<1>::pack(v[u].z - r1.x, v[u].w - r1.y));
                }
            }
        }
    }
    lds_barrier();

    // ---- convolution
    f32x4 acc[kConvSub];
    int base[kConvSub], lpos[kConvSub];
#pragma unroll
    for (int i = 0; i < kConvSub; ++i) {
        acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int r = 16 * (rb0 + i * nw) + (lane & 15);       // output row of this lane in sub-block i (may be >= 80: unused)
        const int a = r / p.lout, l = r - a * p.lout;
        base[i] = a * p.lin;
        lpos[i] = l * p.stride - pad;
    }
    int roff[kConvSub];                                         // LDS element offset of this lane's row under the current tap
    auto tap = [&](int t) {
#pragma unroll
        for (int i = 0; i < kConvSub; ++i) {
            const int li = lpos[i] + t;
            roff[i] = ((li >= 0 && li < p.lin) ? base[i] + li : n_in) * ldk + 8 * kq;
        }
    };
    auto kstep = [&](int kc, const uint4 b0, const uint4 b1) {
#pragma unroll
        for (int i = 0; i < kConvSub; ++i) {
            if (rb0 + i * nw < kConvSub) {                     // wave-uniform
                const int off = roff[i] + 32 * kc;
                const uint4 a_hi = *reinterpret_cast<const uint4 *>(P0 + off);
                const uint4 a_lo = *reinterpret_cast<const uint4 *>(P1 + off);
                f32x4 c = acc[i];                               // smallest terms first; weights first: D^T, 4 channels per lane
                c = Fmt<1>::mfma(b0, a_lo, c);
                c = Fmt<1>::mfma(b1, a_hi, c);
                c = Fmt<1>::mfma(b0, a_hi, c);
                acc[i] = c;
            }
        }
    };
    if constexpr (KS != 0) {
#pragma unroll
        for (int s_ = 0; s_ < NKS; ++s_) {
            if (s_ % NKC == 0) tap(s_ / NKC);
            const uint4 b0 = wh[s_ % kWd], b1 = wl[s_ % kWd];
            if (s_ + kWd < NKS) wfrag(s_ + kWd, wh[s_ % kWd], wl[s_ % kWd]);
            kstep(s_ % NKC, b0, b1);
        }
    } else {
        const int nks = p.ks * nkc;                             // any shape: fragments one K-step ahead
        uint4 nb0, nb1;
        wfrag(0, nb0, nb1);
        for (int t = 0; t < p.ks; ++t) {
            tap(t);
            for (int kc = 0; kc < nkc; ++kc) {
                const uint4 b0 = nb0, b1 = nb1;
                const int sn = t * nkc + kc + 1;
                wfrag(sn < nks ? sn : nks - 1, nb0, nb1);
                kstep(kc, b0, b1);
            }
        }
    }
    lds_barrier();                                              // every wave is done reading the planes
#pragma unroll
    for (int i = 0; i < kConvSub; ++i)
        if (rb0 + i * nw < kConvSub)
            *reinterpret_cast<f32x4 *>(T + (16 * (rb0 + i * nw) + (lane & 15)) * ldt + 16 * cb + 4 * (lane >> 4)) = acc[i];
    lds_barrier();

    // ---- GroupNorm over (lout x cout) per actor, residual, ReLU.  All 512 threads: 512 / na threads per actor (128, 64
    // or 32), each keeps its <= 5 float4 of the actor in registers (its channel quad is the same in every one of them:
    // threads-per-actor is a multiple of cout / 4); the two statistics meet through 32-lane shuffles and one LDS word per
    // half-wave.
    __shared__ float s_red[2][16];
    const int tpa = 512 / na, al = tid / tpa, j = tid - al * tpa;
    const int c4 = p.cout >> 2, n4 = p.lout * c4;               // float4 columns per row, float4s per actor
    const int64_t a = a0 + al;
    const float *Ta = T + al * p.lout * ldt;
    const int c = 4 * (j % c4);
    float4 v[5];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const int i = j + k * tpa;
        v[k] = i < n4 ? *reinterpret_cast<const float4 *>(Ta + (i / c4) * ldt + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        s += (v[k].x + v[k].y) + (v[k].z + v[k].w);
    }
    const float4 g = *reinterpret_cast<const float4 *>(p.gamma + c), bt = *reinterpret_cast<const float4 *>(p.beta + c);
#pragma unroll
    for (int o = 16; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((tid & 31) == 0) s_red[0][tid >> 5] = s;
    lds_barrier();
    const int g0 = (al * tpa) >> 5, ng = tpa >> 5;              // this actor's half-waves
    float mean = 0.f;
    for (int k = 0; k < ng; ++k) mean += s_red[0][g0 + k];
    const float per = (float)(p.lout * p.cout);
    mean = mean / per;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        if (j + k * tpa < n4) {
            const float d0 = v[k].x - mean, d1 = v[k].y - mean, d2 = v[k].z - mean, d3 = v[k].w - mean;
            q += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
        }
    }
#pragma unroll
    for (int o = 16; o >= 1; o >>= 1) q += __shfl_xor(q, o, 64);
    if ((tid & 31) == 0) s_red[1][tid >> 5] = q;
    lds_barrier();
    float var = 0.f;
    for (int k = 0; k < ng; ++k) var += s_red[1][g0 + k];
    const float rstd = 1.0f / sqrtf(var / per + p.eps);
    if (a < p.n_act) {
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const int i = j + k * tpa;
            if (i < n4) {
                const int l = i / c4;
                float4 y = make_float4((v[k].x - mean) * rstd * g.x + bt.x, (v[k].y - mean) * rstd * g.y + bt.y,
                                       (v[k].z - mean) * rstd * g.z + bt.z, (v[k].w - mean) * rstd * g.w + bt.w);
                if (p.res_mode == 1) {
                    const float4 r = *reinterpret_cast<const float4 *>(p.res + (a * p.lout + l) * p.cout + c);
                    y.x += r.x; y.y += r.y; y.z += r.z; y.w += r.w;
                } else if (p.res_mode == 2) {
                    int i0, i1;
                    float w1;
                    const int half = p.lout >> 1;
                    up2_taps(l, half, i0, i1, w1);
                    const float4 r0 = *reinterpret_cast<const float4 *>(p.res + (a * half + i0) * p.cout + c);
                    const float4 r1 = *reinterpret_cast<const float4 *>(p.res + (a * half + i1) * p.cout + c);
                    const float w0 = 1.0f - w1;
                    y.x += w0 * r0.x + w1 * r1.x; y.y += w0 * r0.y + w1 * r1.y;
                    y.z += w0 * r0.z + w1 * r1.z; y.w += w0 * r0.w + w1 * r1.w;
                }
                if (p.relu) { y.x = relu_nan(y.x); y.y = relu_nan(y.y); y.z = relu_nan(y.z); y.w = relu_nan(y.w); }
                *reinterpret_cast<float4 *>(p.out + (a * p.lout + l) * p.cout + c) = y;
            }
        }
    }
}

// ---------------------------------------------------------------- a whole Res1d block in one launch -----
// layers.Res1d (reference layers.py:142-190):  out = relu( GN2(conv2( relu(GN1(conv1(x))) )) + r ),  conv1 k = 3 stride s,
// conv2 k = 3 stride 1, r = x (cin == c, s == 1) or GN_d(conv_d(x)) with conv_d k = 1 stride s.  Same 80-row workgroups:
// the intermediate never leaves the CU -- GN1's output goes straight back into LDS as operand planes (Y), the shortcut's
// 1 x 1 convolution runs on the staged input right behind conv1 and its normalised rows wait in registers.  A second
// block with the identity shortcut can be chained behind the first (an ActorNet group): its input is the first block's
// output as planes, its shortcut the values each thread still holds.
struct Res1dParams {
    const float *x;                    // [A, lin, cin]
    int64_t n_act;
    int lin, cin, c, stride, lout;
    const uint4 *w1, *w2, *wd;         // packed images (wd: null = identity shortcut)
    const float *g1, *b1, *g2, *b2, *gd, *bd;
    const uint4 *w1b, *w2b;            // a second block with the identity shortcut chained behind the first (null: none)
    const float *g1b, *b1b, *g2b, *b2b;
    float eps;
    float *out;                        // [A, lout, c]
};

__global__ __launch_bounds__(512) void k_res1d_gn(const Res1dParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ float s_red[2][16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int na = kConvRows / p.lout;
    const int64_t a0 = (int64_t)blockIdx.x * na;
    const int kpad = conv_kpad(p.cin), ldk = kpad + 8;
    const int n_in = na * p.lin;
    const int ldy = p.c + 8, ldt = p.c + 4;
    // ONE region of LDS serves in turn as the input planes, every fp32 tile and every set of intermediate planes (80 rows
    // + a zero row): each is dead before the next is written -- a tile is consumed into registers by tile_gn (whose two
    // barriers every thread has passed when it returns), planes are done with at the barrier behind their convolution.
    uint16_t *P0 = reinterpret_cast<uint16_t *>(smem), *P1 = P0 + (n_in + 1) * ldk;
    float *T = reinterpret_cast<float *>(smem);
    uint16_t *Y0 = reinterpret_cast<uint16_t *>(smem), *Y1 = Y0 + (kConvRows + 1) * ldy;
    const bool chain = p.w1b != nullptr;

    const int ncb = p.c >> 4, nw = 8 / ncb;
    const int cb = wave % ncb, rb0 = wave / ncb, kq = lane >> 4;
    const bool down = p.wd != nullptr;

    // ---- stage x as two fp16 planes
    {
        const int c4n = kpad / 4, total = (n_in + 1) * c4n;
        for (int i0 = tid; i0 < total; i0 += 4 * 512) {
            float4 v[4];
            int rr[4], cc[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * 512;
                const int r = i / c4n, c = 4 * (i - r * c4n);
                rr[u] = r; cc[u] = c;
                v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                const int ar = r / p.lin;
                const int64_t a = a0 + ar;
                if (i < total && r < n_in && a < p.n_act) {
                    const float *src = p.x + (a * p.lin + (r - ar * p.lin)) * p.cin + c;
                    if (c + 3 < p.cin && (p.cin & 3) == 0) v[u] = *reinterpret_cast<const float4 *>(src);
                    else {
                        if (c < p.cin) v[u].x = src[0];
                        if (c + 1 < p.cin) v[u].y = src[1];
                        if (c + 2 < p.cin) v[u].z = src[2];
                        if (c + 3 < p.cin) v[u].w = src[3];
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (i0 + u * 512 < total) {
                    const uint32_t h0 = Fmt<1>::pack(v[u].x, v[u].y), h1 = Fmt<1>::pack(v[u].z, v[u].w);
                    const f32x2 q0 = Fmt<1>::unpack(h0), q1 = Fmt<1>::unpack(h1);
                    *reinterpret_cast<uint2 *>(P0 + rr[u] * ldk + cc[u]) = make_uint2(h0, h1);
                    *reinterpret_cast<uint2 *>(P1 + rr[u] * ldk + cc[u]) =
                        make_uint2(Fmt<1>::pack(v[u].x - q0.x, v[u].y - q0.y), Fmt<1>::pack(v[u].z - q1.x, v[u].w - q1.y));
                }
            }
        }
    }
    lds_barrier();

    // one convolution as shifted GEMMs over staged planes: out^T = W x^T, weight fragments one K-step ahead
    f32x4 acc[kConvSub];
    auto conv = [&](const uint16_t *Q0, const uint16_t *Q1, int ld, int zero_row, int lin_, int stride_, int ks_, int nkc_,
                    const uint4 *wp) {
        const int pad_ = (ks_ - 1) >> 1;
        int base[kConvSub], lpos[kConvSub];
#pragma unroll
        for (int i = 0; i < kConvSub; ++i) {
            acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            const int r = 16 * (rb0 + i * nw) + (lane & 15);
            const int a = r / p.lout, l = r - a * p.lout;
            base[i] = a * lin_;
            lpos[i] = l * stride_ - pad_;
        }
        auto wfrag = [&](int s_, uint4 &h, uint4 &l_) {
            const int64_t wb = (((int64_t)s_ * ncb + cb) * 2) << 6;
            h = wp[wb + lane];
            l_ = wp[wb + 64 + lane];
        };
        const int nks = ks_ * nkc_;
        uint4 nb0, nb1;
        wfrag(0, nb0, nb1);
        for (int t = 0; t < ks_; ++t) {
            int roff[kConvSub];
#pragma unroll
            for (int i = 0; i < kConvSub; ++i) {
                const int li = lpos[i] + t;
                roff[i] = ((li >= 0 && li < lin_) ? base[i] + li : zero_row) * ld + 8 * kq;
            }
            for (int kc = 0; kc < nkc_; ++kc) {
                const uint4 b0 = nb0, b1 = nb1;
                const int sn = t * nkc_ + kc + 1;
                wfrag(sn < nks ? sn : nks - 1, nb0, nb1);
#pragma unroll
                for (int i = 0; i < kConvSub; ++i) {
                    if (rb0 + i * nw < kConvSub) {
                        const int off = roff[i] + 32 * kc;
                        const uint4 a_hi = *reinterpret_cast<const uint4 *>(Q0 + off);
                        const uint4 a_lo = *reinterpret_cast<const uint4 *>(Q1 + off);
                        f32x4 c = acc[i];
                        c = Fmt<1>::mfma(b0, a_lo, c);
                        c = Fmt<1>::mfma(b1, a_hi, c);
                        c = Fmt<1>::mfma(b0, a_hi, c);
                        acc[i] = c;
                    }
                }
            }
        }
    };
    auto acc_to_tile = [&](const f32x4 (&q)[kConvSub]) {
#pragma unroll
        for (int i = 0; i < kConvSub; ++i)
            if (rb0 + i * nw < kConvSub)
                *reinterpret_cast<f32x4 *>(T + (16 * (rb0 + i * nw) + (lane & 15)) * ldt + 16 * cb + 4 * (lane >> 4)) = q[i];
    };
    // GroupNorm of the tile per actor (512 / na threads each, <= 5 float4 per thread with the same channel quad)
    const int tpa = 512 / na, al = tid / tpa, j = tid - al * tpa;
    const int c4 = p.c >> 2, n4 = p.lout * c4;
    const int64_t a = a0 + al;
    const int c = 4 * (j % c4);
    const float per = (float)(p.lout * p.c);
    const int g0 = (al * tpa) >> 5, ng = tpa >> 5;
    auto tile_gn = [&](float4 (&v)[5], const float *gamma, const float *beta) {     // contains two barriers
        const float4 g = *reinterpret_cast<const float4 *>(gamma + c), bt = *reinterpret_cast<const float4 *>(beta + c);
        const float *Ta = T + al * p.lout * ldt;
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const int i = j + k * tpa;
            v[k] = i < n4 ? *reinterpret_cast<const float4 *>(Ta + (i / c4) * ldt + c) : make_float4(0.f, 0.f, 0.f, 0.f);
            s += (v[k].x + v[k].y) + (v[k].z + v[k].w);
        }
#pragma unroll
        for (int o = 16; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
        if ((tid & 31) == 0) s_red[0][tid >> 5] = s;
        lds_barrier();
        float mean = 0.f;
        for (int k = 0; k < ng; ++k) mean += s_red[0][g0 + k];
        mean = mean / per;
        float q = 0.f;
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            if (j + k * tpa < n4) {
                const float d0 = v[k].x - mean, d1 = v[k].y - mean, d2 = v[k].z - mean, d3 = v[k].w - mean;
                q += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
            }
        }
#pragma unroll
        for (int o = 16; o >= 1; o >>= 1) q += __shfl_xor(q, o, 64);
        if ((tid & 31) == 0) s_red[1][tid >> 5] = q;
        lds_barrier();
        float var = 0.f;
        for (int k = 0; k < ng; ++k) var += s_red[1][g0 + k];
        const float rstd = 1.0f / sqrtf(var / per + p.eps);
#pragma unroll
        for (int k = 0; k < 5; ++k)
            v[k] = make_float4((v[k].x - mean) * rstd * g.x + bt.x, (v[k].y - mean) * rstd * g.y + bt.y,
                               (v[k].z - mean) * rstd * g.z + bt.z, (v[k].w - mean) * rstd * g.w + bt.w);
    };

    // ---- conv1 (and the shortcut's 1 x 1 convolution) on the staged input
    conv(P0, P1, ldk, n_in, p.lin, p.stride, 3, kpad >> 5, p.w1);
    f32x4 acc1[kConvSub];
#pragma unroll
    for (int i = 0; i < kConvSub; ++i) acc1[i] = acc[i];
    if (down) conv(P0, P1, ldk, n_in, p.lin, p.stride, 1, kpad >> 5, p.wd);
    lds_barrier();                                              // the input planes are done with
    acc_to_tile(acc1);
    lds_barrier();
    float4 v[5], res[5];
    tile_gn(v, p.g1, p.b1);
    // a thread's normalised values -> operand planes (row = al * lout + l: the output row numbering)
    auto to_planes = [&](uint16_t *U0, uint16_t *U1, const float4 (&y_)[5]) {
        if (tid < ldy / 4) {                                    // the zero row the padding taps read
            *reinterpret_cast<uint2 *>(U0 + kConvRows * ldy + 4 * tid) = make_uint2(0u, 0u);
            *reinterpret_cast<uint2 *>(U1 + kConvRows * ldy + 4 * tid) = make_uint2(0u, 0u);
        }
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const int i = j + k * tpa;
            if (i < n4) {
                const float4 y = y_[k];
                const int row = al * p.lout + i / c4;
                const uint32_t h0 = Fmt<1>::pack(y.x, y.y), h1 = Fmt<1>::pack(y.z, y.w);
                const f32x2 q0 = Fmt<1>::unpack(h0), q1 = Fmt<1>::unpack(h1);
                *reinterpret_cast<uint2 *>(U0 + row * ldy + c) = make_uint2(h0, h1);
                *reinterpret_cast<uint2 *>(U1 + row * ldy + c) =
                    make_uint2(Fmt<1>::pack(y.x - q0.x, y.y - q0.y), Fmt<1>::pack(y.z - q1.x, y.w - q1.y));
            }
        }
    };
    auto relu5 = [&](float4 (&y_)[5]) {
#pragma unroll
        for (int k = 0; k < 5; ++k)
            y_[k] = make_float4(relu_nan(y_[k].x), relu_nan(y_[k].y), relu_nan(y_[k].z), relu_nan(y_[k].w));
    };
    relu5(v);
    // ---- the shortcut
    if (down) {
        acc_to_tile(acc);                                       // the conv1 tile is in registers everywhere
        lds_barrier();
        tile_gn(res, p.gd, p.bd);
    } else {
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const int i = j + k * tpa;
            res[k] = (a < p.n_act && i < n4) ? *reinterpret_cast<const float4 *>(p.x + (a * p.lout + i / c4) * p.c + c)
                                             : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    to_planes(Y0, Y1, v);                                       // relu(GN1(conv1 x))
    lds_barrier();
    // ---- conv2 on the intermediate, GN2, + shortcut, ReLU
    conv(Y0, Y1, ldy, kConvRows, p.lout, 1, 3, p.c >> 5, p.w2);
    lds_barrier();                                              // the planes are done with: the tile takes their place
    acc_to_tile(acc);
    lds_barrier();
    tile_gn(v, p.g2, p.b2);
    auto add_res_relu = [&]() {
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const bool live = j + k * tpa < n4;
            v[k] = make_float4(live ? relu_nan(v[k].x + res[k].x) : 0.f, live ? relu_nan(v[k].y + res[k].y) : 0.f,
                               live ? relu_nan(v[k].z + res[k].z) : 0.f, live ? relu_nan(v[k].w + res[k].w) : 0.f);
        }
    };
    auto store_out = [&]() {
        if (a < p.n_act) {
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                const int i = j + k * tpa;
                if (i < n4) *reinterpret_cast<float4 *>(p.out + (a * p.lout + i / c4) * p.c + c) = v[k];
            }
        }
    };
    add_res_relu();
    if (!chain) { store_out(); return; }
    // ---- the chained block (identity shortcut = the values this thread holds)
    to_planes(Y0, Y1, v);
#pragma unroll
    for (int k = 0; k < 5; ++k) res[k] = v[k];
    lds_barrier();
    conv(Y0, Y1, ldy, kConvRows, p.lout, 1, 3, p.c >> 5, p.w1b);
    lds_barrier();
    acc_to_tile(acc);
    lds_barrier();
    tile_gn(v, p.g1b, p.b1b);
    relu5(v);
    to_planes(Y0, Y1, v);
    lds_barrier();
    conv(Y0, Y1, ldy, kConvRows, p.lout, 1, 3, p.c >> 5, p.w2b);
    lds_barrier();
    acc_to_tile(acc);
    lds_barrier();
    tile_gn(v, p.g2b, p.b2b);
    add_res_relu();
    store_out();
}

}  // namespace lgcn

using namespace lgcn;

extern "C" {

static bool conv_shape_ok(int cin, int cout, int ks, int stride, int lin, int lout) {
    if (cin < 1 || cin > 128 || (cout != 32 && cout != 64 && cout != 128)) return false;
    if ((ks != 1 && ks != 3) || (stride != 1 && stride != 2) || lin < 1) return false;
    const int pad = (ks - 1) / 2;
    if (lout != (lin + 2 * pad - ks) / stride + 1) return false;
    return lout == 5 || lout == 10 || lout == 20;             // 16 / 8 / 4 actors per workgroup: 512 / na threads each in the GroupNorm phase
}

int64_t lgcn_conv_packed_bytes(int cin, int cout, int ks) {
    if (cin < 1 || cin > 128 || (cout != 32 && cout != 64 && cout != 128) || (ks != 1 && ks != 3)) return LGCN_EINVAL;
    return (int64_t)ks * (conv_kpad(cin) / 32) * (cout / 16) * 2 * 64 * 16;
}

int lgcn_conv_pack_weight(const float *w, int cin, int cout, int ks, void *out, void *stream) {
    if (lgcn_conv_packed_bytes(cin, cout, ks) < 0) return LGCN_EINVAL;
    LGCN_CHECK_PTR(w); LGCN_CHECK_PTR(out); LGCN_CHECK_ALIGN16(out);
    const int64_t total = (int64_t)ks * (conv_kpad(cin) / 32) * (cout / 16) * 64;
    hipLaunchKernelGGL(k_conv_pack, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, cout, cin, ks,
                       reinterpret_cast<uint4 *>(out));
    return launch_status();
}

int lgcn_conv1d_gn(const float *x, int64_t n_act, int lin, int cin, const void *wp, int cout, int ks, int stride,
                   const float *gamma, const float *beta, float eps, const float *res, int res_mode, int relu,
                   float *out, void *stream) {
    if (n_act < 0 || res_mode < 0 || res_mode > 2) return LGCN_EINVAL;
    const int pad = (ks - 1) / 2;
    const int lout = stride > 0 ? (lin + 2 * pad - ks) / stride + 1 : 0;
    if (!conv_shape_ok(cin, cout, ks, stride, lin, lout)) return LGCN_ESHAPE;
    if (res_mode == 2 && (lout & 1)) return LGCN_ESHAPE;
    if (n_act == 0) return LGCN_OK;
    if (n_act > 0x7fffffff / (kConvRows * 128)) return LGCN_ESHAPE;
    const void *al[] = {x, wp, gamma, beta, out};
    for (const void *v : al) { LGCN_CHECK_PTR(v); LGCN_CHECK_ALIGN16(v); }
    if (res_mode != 0) { LGCN_CHECK_PTR(res); LGCN_CHECK_ALIGN16(res); }
    ConvParams p;
    p.x = x; p.n_act = n_act; p.lin = lin; p.cin = cin; p.cout = cout; p.ks = ks; p.stride = stride; p.lout = lout;
    p.wp = reinterpret_cast<const uint4 *>(wp); p.gamma = gamma; p.beta = beta; p.eps = eps;
    p.res = res; p.res_mode = res_mode; p.relu = relu; p.out = out;
    const int na = kConvRows / lout;
    const size_t lds_planes = (size_t)2 * (na * lin + 1) * (conv_kpad(cin) + 8) * 2, lds_tile = (size_t)kConvRows * (cout + 4) * 4;
    const size_t lds = lds_planes > lds_tile ? lds_planes : lds_tile;
    if (lds > 159 * 1024) return LGCN_ESHAPE;                  // the kernel's static words share the 160 KB
    void (*kern)(ConvParams) = k_conv_gn<0, 0>;
    const int nkc = conv_kpad(cin) >> 5;
    if (ks == 1) kern = nkc == 1 ? k_conv_gn<1, 1> : nkc == 2 ? k_conv_gn<1, 2> : nkc == 4 ? k_conv_gn<1, 4> : kern;
    if (ks == 3) kern = nkc == 1 ? k_conv_gn<3, 1> : nkc == 2 ? k_conv_gn<3, 2> : nkc == 4 ? k_conv_gn<3, 4> : kern;
    if (lds > 64 * 1024) {             // above the default ceiling of dynamic LDS (a property set on the code object; idempotent)
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
        if (e != hipSuccess) return (int)e;
    }
    const unsigned grid = (unsigned)((n_act + na - 1) / na);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, (hipStream_t)stream, p);
    return launch_status();
}

static int res1d_launch(const float *x, int64_t n_act, int lin, int cin, int c, int stride, const void *w1p, const float *g1,
                        const float *b1, const void *w2p, const float *g2, const float *b2, const void *wdp, const float *gd,
                        const float *bd, const void *const *second, float eps, float *out, void *stream) {
    if (n_act < 0) return LGCN_EINVAL;
    const int lout = stride > 0 ? (lin + 2 - 3) / stride + 1 : 0;
    if (!conv_shape_ok(cin, c, 3, stride, lin, lout) || (c & 31)) return LGCN_ESHAPE;
    if (wdp == nullptr && (cin != c || stride != 1)) return LGCN_ESHAPE;      // identity shortcut: same shape in and out
    if ((wdp == nullptr) != (gd == nullptr) || (wdp == nullptr) != (bd == nullptr)) return LGCN_EINVAL;
    if (n_act == 0) return LGCN_OK;
    if (n_act > 0x7fffffff / (kConvRows * 128)) return LGCN_ESHAPE;
    const void *al[] = {x, w1p, g1, b1, w2p, g2, b2, out};
    for (const void *v : al) { LGCN_CHECK_PTR(v); LGCN_CHECK_ALIGN16(v); }
    if (wdp != nullptr) { LGCN_CHECK_ALIGN16(wdp); LGCN_CHECK_ALIGN16(gd); LGCN_CHECK_ALIGN16(bd); }
    Res1dParams p;
    p.x = x; p.n_act = n_act; p.lin = lin; p.cin = cin; p.c = c; p.stride = stride; p.lout = lout;
    p.w1 = reinterpret_cast<const uint4 *>(w1p); p.w2 = reinterpret_cast<const uint4 *>(w2p); p.wd = reinterpret_cast<const uint4 *>(wdp);
    p.g1 = g1; p.b1 = b1; p.g2 = g2; p.b2 = b2; p.gd = gd; p.bd = bd; p.eps = eps; p.out = out;
    p.w1b = p.w2b = nullptr; p.g1b = p.b1b = p.g2b = p.b2b = nullptr;
    if (second != nullptr) {          // {w1p, g1, b1, w2p, g2, b2} of the chained block
        for (int i = 0; i < 6; ++i) { LGCN_CHECK_PTR(second[i]); LGCN_CHECK_ALIGN16(second[i]); }
        p.w1b = reinterpret_cast<const uint4 *>(second[0]); p.g1b = reinterpret_cast<const float *>(second[1]);
        p.b1b = reinterpret_cast<const float *>(second[2]); p.w2b = reinterpret_cast<const uint4 *>(second[3]);
        p.g2b = reinterpret_cast<const float *>(second[4]); p.b2b = reinterpret_cast<const float *>(second[5]);
    }
    const int na = kConvRows / lout;
    const size_t in_planes = (size_t)2 * (na * lin + 1) * (conv_kpad(cin) + 8) * 2, tile = (size_t)kConvRows * (c + 4) * 4;
    const size_t mid_planes = (size_t)2 * (kConvRows + 1) * (c + 8) * 2;
    size_t lds = in_planes > tile ? in_planes : tile;          // one region, reused (see the kernel)
    lds = lds > mid_planes ? lds : mid_planes;
    if (lds > 159 * 1024) return LGCN_ESHAPE;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_res1d_gn), hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(k_res1d_gn, dim3((unsigned)((n_act + na - 1) / na)), dim3(512), lds, (hipStream_t)stream, p);
    return launch_status();
}

int lgcn_res1d_gn(const float *x, int64_t n_act, int lin, int cin, int c, int stride, const void *w1p, const float *g1,
                  const float *b1, const void *w2p, const float *g2, const float *b2, const void *wdp, const float *gd,
                  const float *bd, float eps, float *out, void *stream) {
    return res1d_launch(x, n_act, lin, cin, c, stride, w1p, g1, b1, w2p, g2, b2, wdp, gd, bd, nullptr, eps, out, stream);
}

int lgcn_res1d_pair_gn(const float *x, int64_t n_act, int lin, int cin, int c, int stride, const void *w1p, const float *g1,
                       const float *b1, const void *w2p, const float *g2, const float *b2, const void *wdp, const float *gd,
                       const float *bd, const void *w1q, const float *g1q, const float *b1q, const void *w2q, const float *g2q,
                       const float *b2q, float eps, float *out, void *stream) {
    const void *second[6] = {w1q, g1q, b1q, w2q, g2q, b2q};
    return res1d_launch(x, n_act, lin, cin, c, stride, w1p, g1, b1, w2p, g2, b2, wdp, gd, bd, second, eps, out, stream);
}

}  // extern "C"
