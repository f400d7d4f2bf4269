// Floating-point path of the LaneGCN hot path on gfx950 (MI355X):
// fused row-block kernels built from three tile primitives
//   (1) gather-sum of 128-channel rows into a 32 x 128 LDS tile,
//   (2) 32 x 128 x K tile GEMM on v_mfma_f32_32x32x2_f32 (exact fp32 fma chain),
//   (3) per-row GroupNorm(1,128) / ReLU / residual over the LDS tile.
// Workgroup tile = 32 rows x 128 output channels; wave w of the 4 MFMA waves
// owns output channels [32w, 32w+32) and streams its own slice of the packed
// weight straight from L2 to registers (no wave shares a weight element);
// the A operand (gathered rows) is the shared, LDS-resident one.
#include "lgcn_common.hpp"
#include "lgcn_tile.hpp"

namespace lgcn {

constexpr int kTM32 = 32;                  // rows of one f32-MFMA tile (two CSR sub-tiles)
constexpr int kTileFloats = kTM32 * kLDA;  // one 32 x (128+4) LDS tile

// acc[32 x 32 block of this wave] += A[32 x 8*nq] * Wpacked
// A operand of 32x32x2: lane l holds A[l & 31][k = l >> 5]; B operand holds
// B[k = l >> 5][l & 31].  With one float4 per lane per 8 k's, step j of the
// q-th group contracts k = 8q + 4(l >> 5) + j on both operands.
__device__ __forceinline__ void tile_gemm(const float *__restrict__ A, const float4 *__restrict__ wp_wave,
                                          f32x16 &acc, int lane, int nq) {
    const float *arow = A + (lane & 31) * kLDA + 4 * (lane >> 5);
    const float4 *b = wp_wave + lane;
#pragma unroll 4
    for (int q = 0; q < nq; ++q) {
        const float4 a = *reinterpret_cast<const float4 *>(arow + 8 * q);
        const float4 w = b[q * 64];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, w.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, w.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, w.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, w.w, acc, 0, 0, 0);
    }
}

// C/D layout of 32x32: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
__device__ __forceinline__ int acc_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

__device__ __forceinline__ void acc_to_lds(float *T, const f32x16 &acc, int lane, int wave) {
    float *p = T + 32 * wave + (lane & 31);
#pragma unroll
    for (int i = 0; i < 16; ++i) p[acc_row(i, lane) * kLDA] = acc[i];
}

// ------------------------------------------------------------ packing -----
__global__ __launch_bounds__(256) void k_pack_weight(const float *__restrict__ W, int ld, int k_real, int k_pad,
                                                     float *__restrict__ out, int transpose) {
    const int nq = k_pad >> 3;
    const int total = 4 * nq * 64 * 4;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int j = i & 3, lane = (i >> 2) & 63, q = (i >> 8) % nq, w = (i >> 8) / nq;
        const int row = 32 * w + (lane & 31), k = 8 * q + 4 * (lane >> 5) + j;
        out[i] = k < k_real ? (transpose ? W[(int64_t)k * ld + row] : W[(int64_t)row * ld + k]) : 0.f;
    }
}

__global__ __launch_bounds__(256) void k_pack_weight_batch(const lgcn_pack_job_t *__restrict__ jobs) {
    const lgcn_pack_job_t job = jobs[blockIdx.y];
    float *__restrict__ out = reinterpret_cast<float *>(job.out);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;      // < 128 * 128
    const int j = i & 3, lane = (i >> 2) & 63, q = (i >> 8) % (kC >> 3), w = (i >> 8) / (kC >> 3);
    const int row = 32 * w + (lane & 31), k = 8 * q + 4 * (lane >> 5) + j;
    out[i] = job.transpose ? job.W[(int64_t)k * job.ld + row] : job.W[(int64_t)row * job.ld + k];
}

// ------------------------------------------------------------ agg_mlp -----
// 8 waves: waves 0-3 run the MFMA chain of relation i while waves 4-7 gather
// relation i+1 into the other LDS buffer (one barrier per relation).
__device__ __forceinline__ void gather_rel(float *__restrict__ Abuf, const lgcn_agg_mlp_t &p, int ri, int tile,
                                           int gt /*0..255*/) {
    const float4 *__restrict__ src = reinterpret_cast<const float4 *>(p.rel[ri].src);
    const int mode = p.rel[ri].mode;
    const int hw = gt >> 5, l = gt & 31;
    int b[4], e[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int row = it * 8 + hw;
        const int64_t n = (int64_t)tile * kTM32 + row;
        b[it] = 0; e[it] = 0;
        if (n < p.n_rows) {
            if (mode == LGCN_REL_CSR) {   // 32-row tile = CSR sub-tiles 2*tile, 2*tile+1
                const int64_t k = (((int64_t)tile * 2 + (row >> 4)) * p.n_rel_csr + p.rel[ri].ridx) * 16 + (row & 15);
                b[it] = p.rowptr[k]; e[it] = p.rowptr[k + 1];
            } else if (mode == LGCN_REL_RANGE) {
                b[it] = p.rowptr[n]; e[it] = p.rowptr[n + 1];
            } else {
                b[it] = (int)n; e[it] = (int)n + 1;
            }
        }
    }
    float4 s[4];
    // first edge of each of the 4 rows: independent loads in flight together
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        s[it] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (b[it] < e[it]) {
            const int idx = mode == LGCN_REL_CSR ? p.col[b[it]] : b[it];
            s[it] = src[(int64_t)idx * 32 + l];
        }
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        int j = b[it] + 1;
        for (; j + 1 < e[it]; j += 2) {  // two loads in flight, summed in index order
            const int i0 = mode == LGCN_REL_CSR ? p.col[j] : j;
            const int i1 = mode == LGCN_REL_CSR ? p.col[j + 1] : j + 1;
            const float4 x0 = src[(int64_t)i0 * 32 + l];
            const float4 x1 = src[(int64_t)i1 * 32 + l];
            s[it] = f4add(f4add(s[it], x0), x1);
        }
        if (j < e[it]) {
            const int i0 = mode == LGCN_REL_CSR ? p.col[j] : j;
            s[it] = f4add(s[it], src[(int64_t)i0 * 32 + l]);
        }
        *reinterpret_cast<float4 *>(Abuf + (it * 8 + hw) * kLDA + 4 * l) = s[it];
    }
}

// KIND only names the instantiation (1 = LaneConv layer: CSR relations; 0 = every other use) so
// that profilers report the dominant kernel separately; the code is identical.
template <int KIND>
__global__ __launch_bounds__(512) void k_agg_mlp(const lgcn_agg_mlp_t p, int n_tiles) {
    __shared__ __attribute__((aligned(16))) float smem[2 * kTileFloats + 32];
    float *buf0 = smem, *buf1 = smem + kTileFloats;
    int *act = reinterpret_cast<int *>(smem + 2 * kTileFloats);  // [0..15] relation ids, [16] count

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile = xcd_chunk_remap(blockIdx.x, n_tiles);
    const int64_t row0 = (int64_t)tile * kTM32;

    // active relations of this tile (a relation with no edge into the tile
    // contributes an all-zero A tile: skip its gather and its MFMAs)
    if (wave == 0) {
        bool on = false;
        if (lane < p.n_rel) {
            const int mode = p.rel[lane].mode;
            if (mode == LGCN_REL_CSR) {
                const int64_t n_sub = (p.n_rows + 15) >> 4;
                for (int h = 0; h < 2; ++h) {
                    const int64_t sub = (int64_t)tile * 2 + h;
                    if (sub < n_sub) {
                        const int64_t k0 = (sub * p.n_rel_csr + p.rel[lane].ridx) * 16;
                        on = on || p.rowptr[k0 + 16] > p.rowptr[k0];
                    }
                }
            } else if (mode == LGCN_REL_RANGE) {
                const int64_t r1 = row0 + kTM32 < p.n_rows ? row0 + kTM32 : p.n_rows;
                on = p.rowptr[r1] > p.rowptr[row0];
            } else {
                on = true;
            }
        }
        const unsigned long long m = __ballot(on);
        if (on) act[__popcll(m & ((1ull << lane) - 1ull))] = lane;
        if (lane == 0) act[16] = __popcll(m);
    }
    __syncthreads();
    const int nact = __builtin_amdgcn_readfirstlane(act[16]);

    if (wave >= 4 && nact > 0) gather_rel(buf0, p, __builtin_amdgcn_readfirstlane(act[0]), tile, tid - 256);
    __syncthreads();

    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;

    for (int i = 0; i < nact; ++i) {
        float *cur = (i & 1) ? buf1 : buf0;
        float *nxt = (i & 1) ? buf0 : buf1;
        if (wave < 4) {
            const int ri = __builtin_amdgcn_readfirstlane(act[i]);
            tile_gemm(cur, reinterpret_cast<const float4 *>(p.rel[ri].wp) + wave * (16 * 64), acc, lane, 16);
        } else if (i + 1 < nact) {
            gather_rel(nxt, p, __builtin_amdgcn_readfirstlane(act[i + 1]), tile, tid - 256);
        }
        __syncthreads();
    }

    if (wave < 4) {
        if (p.w4 != nullptr) {
            // rank-4 update: the 4 extra input channels of A2M.meta (lanegcn.py:387-395)
            const float4 wc = *reinterpret_cast<const float4 *>(p.w4 + 4 * (32 * wave + (lane & 31)));
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int64_t n = row0 + acc_row(i, lane);
                if (n < p.n_rows) {
                    const float2 tu = reinterpret_cast<const float2 *>(p.x4_a)[n];
                    acc[i] += tu.x * wc.x + tu.y * wc.y + p.x4_b[n] * wc.z + p.x4_c[n] * wc.w;
                }
            }
        }
        acc_to_lds(buf0, acc, lane, wave);
    }
    __syncthreads();

    const int rrow = tid >> 3;  // valid for tid < 256
    const int64_t n = row0 + rrow;
    const bool live = tid < 256 && n < p.n_rows;
    const int flags = p.flags;

    if (!(flags & LGCN_F_GEMM2)) {
        if (tid < 256) {
            RowVals r = row_load(buf0, tid);
            if (live && p.out_pre) row_store_global(p.out_pre + n * kC, tid, r);
            if (flags & LGCN_F_GN1) row_gn(r, tid, p.gn1_g, p.gn1_b, p.eps);
            if (live && (flags & LGCN_F_RES)) row_add_global(r, p.res + n * kC, tid);
            if (flags & LGCN_F_RELU1) row_relu(r);
            if (live) row_store_global(p.out + n * kC, tid, r);
        }
        return;
    }

    if (tid < 256) {
        RowVals r = row_load(buf0, tid);
        if (live && p.out_pre) row_store_global(p.out_pre + n * kC, tid, r);
        if (flags & LGCN_F_GN1) row_gn(r, tid, p.gn1_g, p.gn1_b, p.eps);
        if (flags & LGCN_F_RELU1) row_relu(r);
        if (live && p.out_mid) row_store_global(p.out_mid + n * kC, tid, r);
        row_store_lds(buf0, tid, r);
    }
    __syncthreads();
    if (wave < 4) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        tile_gemm(buf0, reinterpret_cast<const float4 *>(p.wp2) + wave * (16 * 64), acc, lane, 16);
        acc_to_lds(buf1, acc, lane, wave);
    }
    __syncthreads();
    if (tid < 256) {
        RowVals r = row_load(buf1, tid);
        if (live && p.out_pre2) row_store_global(p.out_pre2 + n * kC, tid, r);
        if (flags & LGCN_F_GN2) row_gn(r, tid, p.gn2_g, p.gn2_b, p.eps);
        if (live && (flags & LGCN_F_RES)) row_add_global(r, p.res + n * kC, tid);
        if (flags & LGCN_F_RELU2) row_relu(r);
        if (live) row_store_global(p.out + n * kC, tid, r);
    }
}

// ------------------------------------------------------- mapnet input -----
// h1[row][c] = ReLU(w1[c][0] * x + w1[c][1] * y + b1[c]) for the thread's 16 columns
__device__ __forceinline__ void lin2_relu_to_lds(float *T, int t, float x, float y, const float *__restrict__ w1,
                                                 const float *__restrict__ b1) {
    float *p = T + (t >> 3) * kLDA + 4 * (t & 7);
    const int c0 = 4 * (t & 7);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = c0 + 32 * j;
        const float4 wa = *reinterpret_cast<const float4 *>(w1 + 2 * c);      // (c,0) (c,1) (c+1,0) (c+1,1)
        const float4 wb = *reinterpret_cast<const float4 *>(w1 + 2 * c + 4);  // c+2, c+3
        const float4 bb = *reinterpret_cast<const float4 *>(b1 + c);
        float4 o;
        o.x = relu_nan(x * wa.x + y * wa.y + bb.x);
        o.y = relu_nan(x * wa.z + y * wa.w + bb.y);
        o.z = relu_nan(x * wb.x + y * wb.y + bb.z);
        o.w = relu_nan(x * wb.z + y * wb.w + bb.w);
        *reinterpret_cast<float4 *>(p + 32 * j) = o;
    }
}

__global__ __launch_bounds__(256) void k_mapnet_input(const InputParams p, int n_tiles) {
    __shared__ __attribute__((aligned(16))) float smem[2 * kTileFloats];
    float *T1 = smem, *T2 = smem + kTileFloats;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile = blockIdx.x;
    const int64_t n = (int64_t)tile * kTM32 + (tid >> 3);
    const bool live = n < p.n_rows;
    f32x16 acc;

    float2 c = make_float2(0.f, 0.f), f = make_float2(0.f, 0.f);
    if (live) { c = reinterpret_cast<const float2 *>(p.ctrs)[n]; f = reinterpret_cast<const float2 *>(p.feats)[n]; }

    lin2_relu_to_lds(T1, tid, c.x, c.y, p.wa1, p.ba1);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    tile_gemm(T1, reinterpret_cast<const float4 *>(p.wpa2) + wave * (16 * 64), acc, lane, 16);
    acc_to_lds(T2, acc, lane, wave);
    __syncthreads();
    RowVals ra = row_load(T2, tid);
    row_gn(ra, tid, p.ga, p.bta, p.eps);

    lin2_relu_to_lds(T1, tid, f.x, f.y, p.ws1, p.bs1);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    tile_gemm(T1, reinterpret_cast<const float4 *>(p.wps2) + wave * (16 * 64), acc, lane, 16);
    acc_to_lds(T2, acc, lane, wave);
    __syncthreads();
    RowVals rs = row_load(T2, tid);
    row_gn(rs, tid, p.gs, p.bts, p.eps);
#pragma unroll
    for (int j = 0; j < 4; ++j) rs.v[j] = f4add(rs.v[j], ra.v[j]);
    row_relu(rs);
    if (live) row_store_global(p.out + n * kC, tid, rs);
}

// ---------------------------------------------------------- att pairs -----
__global__ __launch_bounds__(256) void k_att_pairs(const PairParams p) {
    __shared__ __attribute__((aligned(16))) float smem[2 * kTileFloats];
    float *T1 = smem, *T2 = smem + kTileFloats;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int64_t P = *p.n_pairs;
    if (P < 0 || P > p.cap) P = p.cap;
    const int64_t n_tiles = (P + kTM32 - 1) / kTM32;
    f32x16 acc;

    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t pr = tile * kTM32 + (tid >> 3);
        const bool live = pr < P;
        int h = 0, w = 0;
        float dx = 0.f, dy = 0.f;
        if (live) {
            h = p.hi[pr]; w = p.wi[pr];
            const float2 a = reinterpret_cast<const float2 *>(p.agt_ctrs)[h];
            const float2 c = reinterpret_cast<const float2 *>(p.ctx_ctrs)[w];
            dx = a.x - c.x; dy = a.y - c.y;
        }
        lin2_relu_to_lds(T1, tid, dx, dy, p.wd0, p.bd0);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        tile_gemm(T1, reinterpret_cast<const float4 *>(p.wpd2) + wave * (16 * 64), acc, lane, 16);
        acc_to_lds(T2, acc, lane, wave);
        __syncthreads();
        {
            RowVals r = row_load(T2, tid);
            row_gn(r, tid, p.gd, p.btd, p.eps);
            row_relu(r);
            row_store_lds(T2, tid, r);
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        tile_gemm(T2, reinterpret_cast<const float4 *>(p.wpc0e) + wave * (16 * 64), acc, lane, 16);
        acc_to_lds(T1, acc, lane, wave);
        __syncthreads();
        {
            RowVals r = row_load(T1, tid);
            if (live) {
                row_add_global(r, p.U + (int64_t)h * kC, tid);
                row_add_global(r, p.V + (int64_t)w * kC, tid);
            }
            row_gn(r, tid, p.gc, p.btc, p.eps);
            row_relu(r);
            if (live) row_store_global(p.m + pr * kC, tid, r);
        }
        // next iteration's first write to T1 is by the same thread that just
        // read those elements; T2 is rewritten only after the next barrier.
    }
}


// ------------------------------------------------------------- wgrad ------
// dW[r] = dT^T (G_r src_r): block (chunk, r) walks the 32-row tiles chunk, chunk + n_chunks, ... that
// relation r touches; waves 4-7 stage the gathered source rows and the dT rows of the next tile in LDS
// while waves 0-3 contract the current one over its rows on v_mfma_f32_32x32x2_f32 (K-step = 2 rows;
// with lanes along the channel axis both operands are plain row reads, no transpose).  Wave w owns the
// 64 x 64 block (w >> 1, w & 1) of the 128 x 128 result; partials per chunk are summed by k_wgrad_reduce.
__device__ __forceinline__ bool wgrad_tile_active(const lgcn_agg_mlp_t &p, int r, int64_t tile) {
    const int mode = p.rel[r].mode;
    if (mode == LGCN_REL_CSR) {
        const int64_t n_sub = (p.n_rows + 15) >> 4;
        bool on = false;
        for (int h = 0; h < 2; ++h) {
            const int64_t sub = tile * 2 + h;
            if (sub < n_sub) {
                const int64_t k0 = (sub * p.n_rel_csr + p.rel[r].ridx) * 16;
                on = on || p.rowptr[k0 + 16] > p.rowptr[k0];
            }
        }
        return on;
    }
    if (mode == LGCN_REL_RANGE) {
        const int64_t r0 = tile * kTM32, r1 = r0 + kTM32 < p.n_rows ? r0 + kTM32 : p.n_rows;
        return p.rowptr[r1] > p.rowptr[r0];
    }
    return true;
}

__global__ __launch_bounds__(512) void k_wgrad(const lgcn_agg_mlp_t p, const float *__restrict__ dT,
                                               float *__restrict__ part, int n_tiles, int n_chunks) {
    __shared__ __attribute__((aligned(16))) float smem[4 * kTileFloats];
    auto bufA = [&](int b) { return smem + b * kTileFloats; };
    auto bufD = [&](int b) { return smem + (2 + b) * kTileFloats; };
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = blockIdx.y, chunk = blockIdx.x;

    auto next_active = [&](int64_t t) -> int64_t {
        while (t < n_tiles && !wgrad_tile_active(p, r, t)) t += n_chunks;
        return t;
    };
    auto fill = [&](int b, int64_t t) {   // waves 4-7
        const int gt = tid - 256;
        gather_rel(bufA(b), p, r, (int)t, gt);
        const int hw = gt >> 5, l = gt & 31;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int row = it * 8 + hw;
            const int64_t n = t * kTM32 + row;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (n < p.n_rows) v = reinterpret_cast<const float4 *>(dT)[n * 32 + l];
            *reinterpret_cast<float4 *>(bufD(b) + row * kLDA + 4 * l) = v;
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

    int64_t t = next_active(chunk);
    if (wave >= 4 && t < n_tiles) fill(0, t);
    __syncthreads();
    int b = 0;
    const int wj = wave >> 1, wk = wave & 1, i = lane & 31, kk = lane >> 5;
    while (t < n_tiles) {
        const int64_t tn = next_active(t + n_chunks);
        if (wave < 4) {
            const float *D = bufD(b) + kk * kLDA + 64 * wj + i;
            const float *A = bufA(b) + kk * kLDA + 64 * wk + i;
#pragma unroll 4
            for (int s = 0; s < 16; ++s) {
                const float a0 = D[2 * s * kLDA], a1 = D[2 * s * kLDA + 32];
                const float b0 = A[2 * s * kLDA], b1 = A[2 * s * kLDA + 32];
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            }
        } else if (tn < n_tiles) {
            fill(b ^ 1, tn);
        }
        __syncthreads();
        t = tn;
        b ^= 1;
    }
    if (wave < 4) {
        float *o = part + ((int64_t)r * n_chunks + chunk) * (kC * kC);
#pragma unroll
        for (int jb = 0; jb < 2; ++jb)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int g = 0; g < 16; ++g)
                    o[(64 * wj + 32 * jb + acc_row(g, lane)) * kC + 64 * wk + 32 * kb + i] = acc[jb][kb][g];
    }
}

__global__ __launch_bounds__(256) void k_wgrad_reduce(const float *__restrict__ part, int n_chunks, float *__restrict__ dW) {
    const int r = blockIdx.y;
    const int e = blockIdx.x * 256 + threadIdx.x;   // < 128*128
    const float *p = part + (int64_t)r * n_chunks * (kC * kC) + e;
    float s = 0.f;
    for (int c = 0; c < n_chunks; ++c) s += p[(int64_t)c * (kC * kC)];
    dW[(int64_t)r * (kC * kC) + e] = s;
}

}  // namespace lgcn

using namespace lgcn;

extern "C" {

static bool valid_mma(int mma) { return mma >= LGCN_MMA_F32 && mma <= LGCN_MMA_F16X2; }

int64_t lgcn_packed_bytes(int k_pad, int mma) {
    if (!valid_mma(mma) || k_pad < 8 || (k_pad & 7)) return LGCN_EINVAL;
    if (mma == LGCN_MMA_F32) return (int64_t)kC * k_pad * 4;
    if (k_pad != kC) return LGCN_ESHAPE;
    return (int64_t)(mma == LGCN_MMA_BF16X3 ? 3 : mma == LGCN_MMA_F16X2 ? 2 : 1) * kC * kC * 2;
}

int lgcn_pack_weight(const float *W, int ld, int k_real, int k_pad, int mma, void *out, void *stream) {
    LGCN_CHECK_PTR(W); LGCN_CHECK_PTR(out);
    if (!valid_mma(mma) || k_real < 1 || k_pad < k_real || (k_pad & 7) || ld < k_real) return LGCN_EINVAL;
    LGCN_CHECK_ALIGN16(out);
    if (mma != LGCN_MMA_F32) {
        if (k_real != kC || k_pad != kC) return LGCN_ESHAPE;
        return pack_weight_bf(W, ld, mma, 0, out, (hipStream_t)stream);
    }
    const int total = kC * k_pad;
    hipLaunchKernelGGL(k_pack_weight, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, W, ld, k_real, k_pad,
                       reinterpret_cast<float *>(out), 0);
    return launch_status();
}

int lgcn_pack_weight_t(const float *W, int ld, int mma, void *out, void *stream) {
    LGCN_CHECK_PTR(W); LGCN_CHECK_PTR(out);
    if (!valid_mma(mma) || ld < kC) return LGCN_EINVAL;
    LGCN_CHECK_ALIGN16(out);
    if (mma != LGCN_MMA_F32) return pack_weight_bf(W, ld, mma, 1, out, (hipStream_t)stream);
    hipLaunchKernelGGL(k_pack_weight, dim3(kC * kC / 256), dim3(256), 0, (hipStream_t)stream, W, ld, kC, kC,
                       reinterpret_cast<float *>(out), 1);
    return launch_status();
}

int lgcn_pack_weight_batch(const lgcn_pack_job_t *jobs, int n_jobs, int mma, void *stream) {
    if (!valid_mma(mma) || n_jobs < 0 || n_jobs > 65535) return LGCN_EINVAL;
    if (n_jobs == 0) return LGCN_OK;
    LGCN_CHECK_PTR(jobs);
    if (mma != LGCN_MMA_F32) return pack_weight_batch_bf(jobs, n_jobs, mma, (hipStream_t)stream);
    hipLaunchKernelGGL(k_pack_weight_batch, dim3(kC * kC / 256, n_jobs), dim3(256), 0, (hipStream_t)stream, jobs);
    return launch_status();
}

static_assert(sizeof(lgcn_agg_mlp_t) == 32 + LGCN_MAX_REL * 24 + 23 * 8 && LGCN_MAX_REL == 16,
              "lgcn_agg_mlp_t layout: keep lanegcn-1_amd/_lib.py (AggMlp) and tests/test_host_cabi.py in step");

static int validate_agg(const lgcn_agg_mlp_t &p, bool *need_col_out) {
    if (p.n_rows < 0 || p.n_rel < 1 || p.n_rel > LGCN_MAX_REL || !valid_mma(p.mma)) return LGCN_EINVAL;
    constexpr int kKnownFlags = LGCN_F_GN1 | LGCN_F_RELU1 | LGCN_F_GEMM2 | LGCN_F_GN2 | LGCN_F_RES | LGCN_F_RELU2;
#ifdef LGCN_ABLATE
    if (p.flags & ~(kKnownFlags | (1 << 8) | (1 << 9))) return LGCN_EINVAL;
#else
    if (p.flags & ~kKnownFlags) return LGCN_EINVAL;      // unknown bits are an error, not a silent no-op
#endif
    if (p.n_rows == 0) return LGCN_OK;
    if (p.n_rows > 0x7fffffff) return LGCN_ESHAPE;
    LGCN_CHECK_PTR(p.out); LGCN_CHECK_ALIGN16(p.out);
    bool need_rowptr = false, need_col = false;
    for (int r = 0; r < p.n_rel; ++r) {
        LGCN_CHECK_PTR(p.rel[r].src); LGCN_CHECK_PTR(p.rel[r].wp);
        LGCN_CHECK_ALIGN16(p.rel[r].src); LGCN_CHECK_ALIGN16(p.rel[r].wp);
        switch (p.rel[r].mode) {
            case LGCN_REL_IDENT: break;
            case LGCN_REL_CSR:
                if (p.rel[r].ridx < 0 || p.rel[r].ridx >= p.n_rel_csr) return LGCN_EINVAL;
                need_rowptr = need_col = true; break;
            case LGCN_REL_RANGE: need_rowptr = true; break;
            case LGCN_REL_RANGE16:          // split-precision kernels only
                if (p.mma == LGCN_MMA_F32) return LGCN_ESHAPE;
                need_rowptr = true; break;
            default: return LGCN_EINVAL;
        }
    }
    if (need_rowptr) LGCN_CHECK_PTR(p.rowptr);
    if (need_col) LGCN_CHECK_PTR(p.col);
    if (need_col) {   // one rowptr per launch: a CSR plan and a RANGE prefix cannot be mixed
        for (int r = 0; r < p.n_rel; ++r)
            if (p.rel[r].mode == LGCN_REL_RANGE || p.rel[r].mode == LGCN_REL_RANGE16) return LGCN_EINVAL;
    }
    if (p.flags & LGCN_F_GN1) { LGCN_CHECK_PTR(p.gn1_g); LGCN_CHECK_PTR(p.gn1_b); LGCN_CHECK_ALIGN16(p.gn1_g); LGCN_CHECK_ALIGN16(p.gn1_b); }
    if (p.flags & LGCN_F_GEMM2) { LGCN_CHECK_PTR(p.wp2); LGCN_CHECK_ALIGN16(p.wp2); }
    if (p.flags & LGCN_F_GN2) {
        if (!(p.flags & LGCN_F_GEMM2)) return LGCN_EINVAL;
        LGCN_CHECK_PTR(p.gn2_g); LGCN_CHECK_PTR(p.gn2_b); LGCN_CHECK_ALIGN16(p.gn2_g); LGCN_CHECK_ALIGN16(p.gn2_b);
    }
    if ((p.flags & LGCN_F_RELU2) && !(p.flags & LGCN_F_GEMM2)) return LGCN_EINVAL;
    if (p.flags & LGCN_F_RES) { LGCN_CHECK_PTR(p.res); LGCN_CHECK_ALIGN16(p.res); }
    if (p.w4) { LGCN_CHECK_PTR(p.x4_a); LGCN_CHECK_PTR(p.x4_b); LGCN_CHECK_PTR(p.x4_c); LGCN_CHECK_ALIGN16(p.w4); }
    if (p.out_pre) LGCN_CHECK_ALIGN16(p.out_pre);
    if (p.out_mid) LGCN_CHECK_ALIGN16(p.out_mid);
    if (p.out_pre2) LGCN_CHECK_ALIGN16(p.out_pre2);
    if (p.ch_wu) {
        const void *q[] = {p.ch_wq, p.ch_gq_g, p.ch_gq_b, p.ch_wu, p.ch_u_out};
        for (const void *v : q) { LGCN_CHECK_PTR(v); LGCN_CHECK_ALIGN16(v); }
    }
    if (p.ch_wv) { LGCN_CHECK_PTR(p.ch_v_out); LGCN_CHECK_ALIGN16(p.ch_wv); LGCN_CHECK_ALIGN16(p.ch_v_out); }
    if ((p.ch_wu || p.ch_wv) && need_col) return LGCN_EINVAL;      // chained outputs: row blocks without CSR relations
    *need_col_out = need_col;
    return LGCN_OK;
}

// The chained outputs of a block as launches of their own on its `out` rows (exact-f32 mode, whose kernels have no
// chained stages): the same arithmetic, lanegcn.py:696-699.
static int chain_as_launches(const lgcn_agg_mlp_t &p, void *stream) {
    lgcn_agg_mlp_t q{};
    q.n_rows = p.n_rows; q.n_rel = 1; q.eps = p.eps; q.mma = p.mma;
    q.rel[0].src = p.out; q.rel[0].mode = LGCN_REL_IDENT;
    if (p.ch_wu) {
        q.rel[0].wp = p.ch_wq; q.flags = LGCN_F_GN1 | LGCN_F_RELU1 | LGCN_F_GEMM2;
        q.gn1_g = p.ch_gq_g; q.gn1_b = p.ch_gq_b; q.wp2 = p.ch_wu; q.out = p.ch_u_out;
        const int rc = lgcn_agg_mlp(&q, stream);
        if (rc != LGCN_OK) return rc;
    }
    if (p.ch_wv) {
        q.rel[0].wp = p.ch_wv; q.flags = 0; q.gn1_g = q.gn1_b = q.wp2 = nullptr; q.out = p.ch_v_out;
        return lgcn_agg_mlp(&q, stream);
    }
    return LGCN_OK;
}

int lgcn_agg_mlp(const lgcn_agg_mlp_t *ph, void *stream) {
    LGCN_CHECK_PTR(ph);
    const lgcn_agg_mlp_t &p = *ph;
    bool need_col = false;
    const int rc = validate_agg(p, &need_col);
    if (rc != LGCN_OK || p.n_rows == 0) return rc;
    if (p.mma != LGCN_MMA_F32) return agg_mlp_bf(p, need_col, (hipStream_t)stream);
    const int n_tiles = (int)((p.n_rows + kTM32 - 1) / kTM32);
    if (need_col)
        hipLaunchKernelGGL((k_agg_mlp<1>), dim3(n_tiles), dim3(512), 0, (hipStream_t)stream, p, n_tiles);
    else
        hipLaunchKernelGGL((k_agg_mlp<0>), dim3(n_tiles), dim3(512), 0, (hipStream_t)stream, p, n_tiles);
    const int st = launch_status();
    return st != LGCN_OK || !(p.ch_wu || p.ch_wv) ? st : chain_as_launches(p, stream);
}

int lgcn_agg_mlp_multi(const lgcn_agg_mlp_t *const *ps, int n, void *stream) {
    LGCN_CHECK_PTR(ps);
    if (n < 1 || n > LGCN_MAX_MULTI) return LGCN_EINVAL;
    bool one = true;
    for (int i = 0; i < n; ++i) {
        LGCN_CHECK_PTR(ps[i]);
        bool c = false;
        const int rc = validate_agg(*ps[i], &c);
        if (rc != LGCN_OK) return rc;
        one = one && ps[i]->n_rows > 0 && ps[i]->mma == ps[0]->mma && ps[i]->mma != LGCN_MMA_F32 && !c && ps[i]->tile_rb == ps[0]->tile_rb;
    }
    if (one && n > 1) return agg_mlp_multi_bf(ps, n, (hipStream_t)stream);
    for (int i = 0; i < n; ++i) {
        const int rc = lgcn_agg_mlp(ps[i], stream);
        if (rc != LGCN_OK) return rc;
    }
    return LGCN_OK;
}

int lgcn_agg_mlp_pair(const lgcn_agg_mlp_t *a, const lgcn_agg_mlp_t *b, void *stream) {
    LGCN_CHECK_PTR(a); LGCN_CHECK_PTR(b);
    bool ca = false, cb = false;
    int rc = validate_agg(*a, &ca);
    if (rc != LGCN_OK) return rc;
    rc = validate_agg(*b, &cb);
    if (rc != LGCN_OK) return rc;
    // one launch only for two non-empty split-precision problems without CSR relations in the same mode
    if (a->n_rows > 0 && b->n_rows > 0 && a->mma == b->mma && a->mma != LGCN_MMA_F32 && !ca && !cb &&
        a->tile_rb == 0 && b->tile_rb == 0)
        return agg_mlp_pair_bf(*a, *b, (hipStream_t)stream);
    rc = lgcn_agg_mlp(a, stream);
    return rc != LGCN_OK ? rc : lgcn_agg_mlp(b, stream);
}

int lgcn_mapnet_input(const float *ctrs, const float *feats, int64_t n_rows, const float *wa1, const float *ba1,
                      const float *wpa2, const float *ga, const float *bta, const float *ws1, const float *bs1,
                      const float *wps2, const float *gs, const float *bts, float eps, int mma, float *out, void *stream) {
    if (n_rows < 0 || !valid_mma(mma)) return LGCN_EINVAL;
    if (n_rows == 0) return LGCN_OK;
    if (n_rows > 0x7fffffff) return LGCN_ESHAPE;
    const void *ptrs[] = {ctrs, feats, wa1, ba1, wpa2, ga, bta, ws1, bs1, wps2, gs, bts, out};
    for (const void *q : ptrs) LGCN_CHECK_PTR(q);
    const void *al[] = {wa1, ba1, wpa2, ga, bta, ws1, bs1, wps2, gs, bts, out};
    for (const void *q : al) LGCN_CHECK_ALIGN16(q);
    InputParams p{ctrs, feats, n_rows, wa1, ba1, wpa2, ga, bta, ws1, bs1, wps2, gs, bts, eps, out};
    if (mma != LGCN_MMA_F32) return mapnet_input_bf(p, mma, (hipStream_t)stream);
    const int n_tiles = (int)((n_rows + kTM32 - 1) / kTM32);
    hipLaunchKernelGGL(k_mapnet_input, dim3(n_tiles), dim3(256), 0, (hipStream_t)stream, p, n_tiles);
    return launch_status();
}

int lgcn_att_pairs(const float *agt_ctrs, const float *ctx_ctrs, const int32_t *hi, const int32_t *wi,
                   const int32_t *n_pairs, int64_t cap, const float *wd0, const float *bd0, const float *wpd2,
                   const float *gd, const float *btd, const float *wpc0e, const float *U, const float *V,
                   const float *gc, const float *btc, float eps, int mma, float *m, void *stream) {
    if (cap < 0 || !valid_mma(mma)) return LGCN_EINVAL;
    if (cap == 0) return LGCN_OK;
    if (cap > 0x7ffffff0) return LGCN_ESHAPE;
    const void *ptrs[] = {agt_ctrs, ctx_ctrs, hi, wi, n_pairs, wd0, bd0, wpd2, gd, btd, wpc0e, U, V, gc, btc, m};
    for (const void *q : ptrs) LGCN_CHECK_PTR(q);
    const void *al[] = {wd0, bd0, wpd2, gd, btd, wpc0e, U, V, gc, btc, m};
    for (const void *q : al) LGCN_CHECK_ALIGN16(q);
    PairParams p{agt_ctrs, ctx_ctrs, hi, wi, n_pairs, cap, wd0, bd0, wpd2, gd, btd, wpc0e, U, V, gc, btc, eps, m};
    if (mma != LGCN_MMA_F32) return att_pairs_bf(p, mma, (hipStream_t)stream);
    int64_t tiles = (cap + kTM32 - 1) / kTM32;
    const unsigned grid = (unsigned)(tiles < 2048 ? tiles : 2048);
    hipLaunchKernelGGL(k_att_pairs, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
    return launch_status();
}

int lgcn_wgrad(const lgcn_agg_mlp_t *ph, const float *dT, float *dW, float *part, int n_chunks, void *stream) {
    LGCN_CHECK_PTR(ph); LGCN_CHECK_PTR(dT); LGCN_CHECK_PTR(dW); LGCN_CHECK_PTR(part);
    const lgcn_agg_mlp_t &p = *ph;
    if (p.n_rows < 0 || p.n_rel < 1 || p.n_rel > LGCN_MAX_REL || n_chunks < 1 || n_chunks > 1024) return LGCN_EINVAL;
    if (p.n_rows > 0x7fffffff) return LGCN_ESHAPE;
    LGCN_CHECK_ALIGN16(dT); LGCN_CHECK_ALIGN16(dW); LGCN_CHECK_ALIGN16(part);
    bool need_rowptr = false, need_col = false;
    for (int r = 0; r < p.n_rel; ++r) {
        LGCN_CHECK_PTR(p.rel[r].src); LGCN_CHECK_ALIGN16(p.rel[r].src);
        switch (p.rel[r].mode) {
            case LGCN_REL_IDENT: break;
            case LGCN_REL_CSR:
                if (p.rel[r].ridx < 0 || p.rel[r].ridx >= p.n_rel_csr) return LGCN_EINVAL;
                need_rowptr = need_col = true; break;
            case LGCN_REL_RANGE: need_rowptr = true; break;
            default: return LGCN_EINVAL;
        }
    }
    if (need_rowptr) LGCN_CHECK_PTR(p.rowptr);
    if (need_col) LGCN_CHECK_PTR(p.col);
    hipStream_t st = (hipStream_t)stream;
    const int n_tiles = (int)((p.n_rows + kTM32 - 1) / kTM32);
    hipLaunchKernelGGL(k_wgrad, dim3(n_chunks, p.n_rel), dim3(512), 0, st, p, dT, part, n_tiles, n_chunks);
    hipLaunchKernelGGL(k_wgrad_reduce, dim3(kC * kC / 256, p.n_rel), dim3(256), 0, st, part, n_chunks, dW);
    return launch_status();
}

}  // extern "C"
