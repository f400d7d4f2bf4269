// Integer path of the LaneGCN hot path: graph_gather offsets, tile-major CSR
// plan, distance-gated pair search.  Everything here is bit-exact against the
// reference (lanegcn.py:171-209, 672-689); this file is compiled with
// -ffp-contract=off so the pair-search distance is evaluated exactly like
// ATen's sub / pow(2) / sum / sqrt / le chain (no FMA).
#include "lgcn_common.hpp"

namespace lgcn {

// ---------------------------------------------------------------- scan ----
// Exclusive scan of int32: per-block scan + block totals, then an add-back in which every block sums the totals
// before it (two launches; beyond 4096 blocks the totals are scanned by their own single-block launch first).
constexpr int kScanThreads = 256;
constexpr int kScanItems = 8;
constexpr int kScanTile = kScanThreads * kScanItems;

// Inclusive scan over the 64 lanes at VALU speed: Hillis-Steele inside every row of 16 lanes (DPP row_shr 1, 2, 4, 8: a lane
// without a source lane adds 0), then the last lane of row 0 / 2 into rows 1 / 3 (row_bcast:15) and lane 31 into rows 2
// and 3 (row_bcast:31).  (__shfl_up is ds_bpermute: an LDS round trip per step.)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_add_from(int v) {
    return v + __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xf, false);
}
__device__ __forceinline__ int wave_inclusive_scan(int v, int /*lane*/) {
    v = dpp_add_from<0x111, 0xf>(v);
    v = dpp_add_from<0x112, 0xf>(v);
    v = dpp_add_from<0x114, 0xf>(v);
    v = dpp_add_from<0x118, 0xf>(v);
    v = dpp_add_from<0x142, 0xa>(v);
    v = dpp_add_from<0x143, 0xc>(v);
    return v;
}

// Block-wide exclusive scan of one value per thread; returns the exclusive
// prefix, *total = block sum.  NT threads (multiple of 64, <= 1024).
template <int NT>
__device__ __forceinline__ int block_exclusive_scan(int v, int *total, int *lds /*[NT/64 + 1]*/) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int inc = wave_inclusive_scan(v, lane);
    if (lane == 63) lds[w] = inc;
    __syncthreads();
    if (w == 0) {      // the wave totals are scanned by the first wave (it was a serial loop of one thread: 16 LDS round trips)
        const int t = lane < NT / 64 ? lds[lane] : 0;
        const int run = wave_inclusive_scan(t, lane);
        if (lane < NT / 64) lds[lane] = run - t;
        if (lane == NT / 64 - 1) lds[NT / 64] = run;
    }
    __syncthreads();
    const int base = lds[w];
    *total = lds[NT / 64];
    __syncthreads();
    return base + inc - v;
}

__global__ __launch_bounds__(kScanThreads) void k_scan_blocks(const int32_t *in, int32_t *out,
                                                              int32_t *sums, int64_t n) {
    __shared__ int lds[kScanThreads / 64 + 1];
    const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
    int v[kScanItems];
    int s = 0;
#pragma unroll
    for (int i = 0; i < kScanItems; ++i) {
        v[i] = (base + i < n) ? in[base + i] : 0;
        s += v[i];
    }
    int total;
    int pre = block_exclusive_scan<kScanThreads>(s, &total, lds);
#pragma unroll
    for (int i = 0; i < kScanItems; ++i) {
        if (base + i < n) out[base + i] = pre;
        pre += v[i];
    }
    if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

__global__ __launch_bounds__(1024) void k_scan_sums(int32_t *sums, int nb) {
    __shared__ int lds[1024 / 64 + 1];
    int carry = 0;
    for (int c = 0; c < nb; c += 1024) {
        const int i = c + threadIdx.x;
        const int v = i < nb ? sums[i] : 0;
        int total;
        const int pre = block_exclusive_scan<1024>(v, &total, lds);
        if (i < nb) sums[i] = pre + carry;
        carry += total;
    }
}

// Add-back with the scan of the block totals folded in: block b sums the totals of the blocks before it (a few
// hundred at most at the plan's sizes) instead of waiting for a separate single-block scan launch.
__global__ __launch_bounds__(kScanThreads) void k_scan_add_prefix(int32_t *out, const int32_t *sums, int64_t n) {
    __shared__ int lds[kScanThreads / 64 + 1];
    int part = 0;
    for (int i = threadIdx.x; i < (int)blockIdx.x; i += kScanThreads) part += sums[i];
    int add;
    block_exclusive_scan<kScanThreads>(part, &add, lds);
    const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
#pragma unroll
    for (int i = 0; i < kScanItems; ++i)
        if (base + i < n) out[base + i] += add;
}

__global__ __launch_bounds__(kScanThreads) void k_scan_add(int32_t *out, const int32_t *sums, int64_t n) {
    const int add = sums[blockIdx.x];
    const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
#pragma unroll
    for (int i = 0; i < kScanItems; ++i)
        if (base + i < n) out[base + i] += add;
}

inline int64_t scan_ws_elems(int64_t n) { return (n + kScanTile - 1) / kScanTile + 1; }

// out may alias in.
static int exclusive_scan(const int32_t *in, int32_t *out, int64_t n, int32_t *sums, hipStream_t st) {
    if (n <= 0) return LGCN_OK;
    const int64_t nb = (n + kScanTile - 1) / kScanTile;
    if (nb > 0x7fffffff) return LGCN_ESHAPE;
    hipLaunchKernelGGL(k_scan_blocks, dim3((unsigned)nb), dim3(kScanThreads), 0, st, in, out, sums, n);
    if (nb > 4096) {          // very long inputs: scan the block totals in their own launch
        hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(1024), 0, st, sums, (int)nb);
        hipLaunchKernelGGL(k_scan_add, dim3((unsigned)nb), dim3(kScanThreads), 0, st, out, sums, n);
    } else if (nb > 1) {
        hipLaunchKernelGGL(k_scan_add_prefix, dim3((unsigned)nb), dim3(kScanThreads), 0, st, out, sums, n);
    }
    return launch_status();
}

// -------------------------------------------------------- graph_gather ----
__global__ __launch_bounds__(256) void k_graph_gather(const int64_t *in, int64_t n,
                                                      const int64_t *seg_off, const int64_t *seg_base,
                                                      int n_seg, int64_t *out64, int32_t *out32) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n;
         e += (int64_t)gridDim.x * blockDim.x) {
        // last segment with seg_off[s] <= e (empty segments are skipped by
        // taking the LAST one: seg_off is non-decreasing)
        int lo = 0, hi = n_seg;  // invariant: seg_off[lo] <= e < seg_off[hi]
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (seg_off[mid] <= e) lo = mid; else hi = mid;
        }
        const int64_t v = in[e] + seg_base[lo];
        if (out64) out64[e] = v;
        if (out32) out32[e] = (int32_t)v;
    }
}

// ----------------------------------------------------------- CSR plan -----
struct CooTable {
    const int64_t *u[LGCN_MAX_REL];
    const int64_t *v[LGCN_MAX_REL];
    int64_t start[LGCN_MAX_REL + 1];  // prefix of n_edges
    int n_rel;
    int64_t n_nodes;
};

__device__ __forceinline__ int64_t csr_key(int64_t n, int r, int n_rel) {
    return ((n >> 4) * n_rel + r) * 16 + (n & 15);
}

// pass 0: count, pass 1: fill
template <int PASS>
__global__ __launch_bounds__(256) void k_csr_edges(const CooTable t, int32_t *cnt_or_cursor,
                                                   const int32_t *rowptr, int32_t *col) {
    const int64_t total = t.start[t.n_rel];
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (int64_t)gridDim.x * blockDim.x) {
        int r = 0;
        while (r + 1 < t.n_rel && e >= t.start[r + 1]) ++r;
        const int64_t le = e - t.start[r];
        const int64_t u = t.u[r][le], v = t.v[r][le];
        if (u < 0 || u >= t.n_nodes || v < 0 || v >= t.n_nodes) continue;  // never index out of bounds
        const int64_t k = csr_key(u, r, t.n_rel);
        if (PASS == 0) {
            atomicAdd(&cnt_or_cursor[k], 1);
        } else {
            const int pos = rowptr[k] + atomicAdd(&cnt_or_cursor[k], 1);
            col[pos] = (int32_t)v;
        }
    }
}

// Canonical order inside each row (ascending source index) so that the
// floating-point gather-sum that consumes the plan is deterministic.
__global__ __launch_bounds__(256) void k_csr_sort_rows(const int32_t *rowptr, int32_t *col, int64_t n_keys) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_keys) return;
    const int b = rowptr[k], e = rowptr[k + 1];
    for (int i = b + 1; i < e; ++i) {
        const int x = col[i];
        int j = i - 1;
        while (j >= b && col[j] > x) { col[j + 1] = col[j]; --j; }
        col[j + 1] = x;
    }
}

// -------------------------------------------------------- pair search -----
__device__ __forceinline__ int find_scene(const int32_t *off, int n_scenes, int g) {
    int lo = 0, hi = n_scenes;  // off[lo] <= g < off[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (off[mid] <= g) lo = mid; else hi = mid;
    }
    return lo;
}

// lanegcn.py:676-678: dist = a - c; sqrt((dist ** 2).sum(2)) <= th, fp32, no FMA.
__device__ __forceinline__ bool within(float ax, float ay, float cx, float cy, float th) {
    const float dx = __fsub_rn(ax, cx), dy = __fsub_rn(ay, cy);
    const float d2 = __fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy));
    return __fsqrt_rn(d2) <= th;
}

// One wave per target row.  PASS 0 counts, PASS 1 writes (hi, wi) and the segment table of the later
// index_add_(0, hi, .):  rowptr_q[h] = first p with hi[p] >= h.  hi is non-decreasing and a row g of a scene that
// advances the numbering owns h = hval(g), so rowptr_q[hval(g)] = rowptr[g]; rows of scenes that do not advance
// it (legacy: scenes without pairs) fill the tail [h_used, T), where no pair can follow: P.  rowptr_q[T] = P.
template <int PASS>
__device__ __forceinline__ void pairs_rows_body(const float2 *agt, const int32_t *agt_off,
                                                const float2 *ctx, const int32_t *ctx_off,
                                                int n_scenes, int n_agt, int n_ctx, float th,
                                                int32_t *rowcnt, const int32_t *rowptr,
                                                const int32_t *hi_base, const int32_t *wi_base,
                                                int32_t *hi, int32_t *wi, int64_t cap, int legacy,
                                                int32_t *rowptr_q, int row_block) {
    const int lane = threadIdx.x & 63;
    const int g = row_block * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (g >= n_agt) return;
    const int sc = find_scene(agt_off, n_scenes, g);
    // offset tables are caller data: a table that does not describe this job (stale memory under a captured graph)
    // must not take a load or a store out of bounds
    int c0 = ctx_off[sc], c1 = ctx_off[sc + 1];
    c0 = c0 < 0 ? 0 : c0 > n_ctx ? n_ctx : c0;
    c1 = c1 < c0 ? c0 : c1 > n_ctx ? n_ctx : c1;
    const float2 a = agt[g];
    int count = 0;
    int64_t pos = 0;
    int hval = 0, wbase = 0;
    if (PASS == 1) {
        pos = rowptr[g];
        hval = g - agt_off[sc] + hi_base[sc];
        wbase = wi_base[sc] - c0;
        if (lane == 0) {
            const int P = rowptr[n_agt];
            int s0 = agt_off[sc], s1 = agt_off[sc + 1];
            s0 = s0 < 0 ? 0 : s0 > n_agt ? n_agt : s0;
            s1 = s1 < s0 ? s0 : s1 > n_agt ? n_agt : s1;
            const bool advances = !legacy || rowptr[s1] - rowptr[s0] > 0;
            const int q2 = hi_base[n_scenes] + g - hi_base[sc];
            // a capacity below the true pair count (n_pairs comes back negative): the segment table describes the pairs
            // that were kept, [0, cap) -- whatever walks it afterwards stays inside the caller's [cap, .] buffers
            const int Pc = (int64_t)P > cap ? (int)cap : P;
            if (advances) { if (hval >= 0 && hval <= n_agt) rowptr_q[hval] = pos > cap ? (int)cap : (int)pos; }
            else if (q2 >= 0 && q2 <= n_agt) rowptr_q[q2] = Pc;
            if (g == 0) rowptr_q[n_agt] = Pc;
        }
    }
    for (int s0 = c0; s0 < c1; s0 += 64) {
        const int s = s0 + lane;
        bool ok = false;
        if (s < c1) {
            const float2 c = ctx[s];
            ok = within(a.x, a.y, c.x, c.y, th);
        }
        const unsigned long long m = __ballot(ok);
        if (PASS == 0) {
            count += __popcll(m);
        } else {
            if (ok) {
                const int64_t o = pos + __popcll(m & ((1ull << lane) - 1ull));
                if (o < cap) { hi[o] = hval; wi[o] = s + wbase; }
            }
            pos += __popcll(m);
        }
    }
    if (PASS == 0 && lane == 0) rowcnt[g] = count;
}

// Workspace layout of one pair search: rowptr_true [T+1] | hi_base [B+1] | wi_base [B]
struct PairsJob {
    const float2 *agt; const int32_t *agt_off;
    const float2 *ctx; const int32_t *ctx_off;
    int n_scenes, n_agt, n_ctx, legacy;
    float th;
    int32_t *hi, *wi;
    int64_t cap;
    int32_t *n_pairs, *rowptr_q, *ws;
    __device__ int32_t *rp() const { return ws; }
    __device__ int32_t *hi_base() const { return ws + (n_agt + 1); }
    __device__ int32_t *wi_base() const { return ws + (n_agt + 1) + n_scenes + 1; }
};
struct PairsJobs { PairsJob j[4]; };

// blockIdx.y selects the job; blocks past a job's rows return at once
template <int PASS>
__global__ __launch_bounds__(256) void k_pairs_rows(const PairsJobs jobs) {
    const PairsJob &j = jobs.j[blockIdx.y];
    pairs_rows_body<PASS>(j.agt, j.agt_off, j.ctx, j.ctx_off, j.n_scenes, j.n_agt, j.n_ctx, j.th, j.rp(), j.rp(), j.hi_base(),
                          j.wi_base(), j.hi, j.wi, j.cap, j.legacy, j.rowptr_q, (int)blockIdx.x);
}

// Single block: exclusive scan of the row counts in place (rp[0..T), rp[T] = P; each thread owns a contiguous
// run), then the per-scene index bases (lanegcn.py:681-687; legacy: a scene without pairs does not advance the
// running counts), hi_base[n_scenes] = rows numbered in all (h_used), and P.  The pair search has at most a few
// 10^4 rows, so one block replaces four launches.
__device__ __forceinline__ void pairs_scan_bases_body(int32_t *rp, int n_agt, const int32_t *agt_off,
                                                      const int32_t *ctx_off, int n_scenes, int legacy, int64_t cap,
                                                      int32_t *hi_base, int32_t *wi_base, int32_t *n_pairs) {
    __shared__ int lds[1024 / 64 + 1];
    // chunks of 4096 counts: every thread scans 4 consecutive ones (adjacent threads touch adjacent words)
    int tot = 0;
    for (int64_t c0 = 0; c0 <= n_agt; c0 += 4096) {
        const int64_t b = c0 + 4 * (int64_t)threadIdx.x;
        int v[4], sum = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            v[k] = b + k < n_agt ? rp[b + k] : 0;      // rp[T] holds no count
            sum += v[k];
        }
        int chunk;
        int off = block_exclusive_scan<1024>(sum, &chunk, lds) + tot;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (b + k <= n_agt) rp[b + k] = off;
            off += v[k];
        }
        tot += chunk;
    }
    __syncthreads();    // the scanned table is read across threads below
    int carry_h = 0, carry_w = 0;
    for (int c = 0; c < n_scenes; c += 1024) {
        const int i = c + threadIdx.x;
        int th = 0, tw = 0;
        if (i < n_scenes) {
            int a0 = agt_off[i], a1 = agt_off[i + 1];
            a0 = a0 < 0 ? 0 : a0 > n_agt ? n_agt : a0;          // caller data: never index rp[] out of bounds
            a1 = a1 < a0 ? a0 : a1 > n_agt ? n_agt : a1;
            const bool nonempty = rp[a1] - rp[a0] > 0;
            if (!legacy || nonempty) { th = a1 - a0; tw = ctx_off[i + 1] - ctx_off[i]; }
        }
        int tot_h, tot_w;
        const int ph = block_exclusive_scan<1024>(th, &tot_h, lds);
        const int pw = block_exclusive_scan<1024>(tw, &tot_w, lds);
        if (i < n_scenes) { hi_base[i] = ph + carry_h; wi_base[i] = pw + carry_w; }
        carry_h += tot_h;
        carry_w += tot_w;
    }
    if (threadIdx.x == 0) {
        hi_base[n_scenes] = carry_h;
        *n_pairs = (int64_t)tot > cap ? -tot : tot;
    }
}

__global__ __launch_bounds__(1024) void k_pairs_scan_bases(const PairsJobs jobs) {    // one block per job
    const PairsJob &j = jobs.j[blockIdx.x];
    pairs_scan_bases_body(j.rp(), j.n_agt, j.agt_off, j.ctx_off, j.n_scenes, j.legacy, j.cap, j.hi_base(), j.wi_base(),
                          j.n_pairs);
    if (j.n_agt == 0 && threadIdx.x == 0) j.rowptr_q[0] = 0;     // no rows: the fill pass never runs for this job
}

// ------------------------------------------------- fused index pipeline -----
// lgcn_index_build: graph_gather + CSR plan + up to four pair searches in FOUR launches (count | scan | fill | sort)
// instead of twelve.  At S2 every one of those launches is a few microseconds of work behind a launch boundary
// of the same size; what the twelve cost is their number.
//   * graph_gather is folded into the edge pass: an edge thread turns its two local indices into global ones
//     (binary search of the segment table) and keeps them as an int32 pair for the fill pass.
//   * the per-key counter word serves the count pass (low half: edges of the key) AND the fill pass (high half: slots
//     handed out); the sort pass, which puts each key's entries in canonical (ascending source) order, writes it
//     back to ZERO.  The counters must be zero on entry and are zero again on completion: no zeroing launch and no
//     separate cursor array.  (Sorting a key from the thread that fills its last slot -- one launch fewer -- was
//     measured 8 x slower: the release/acquire pair it needs is a write-back + invalidate of an XCD's L2 per wave.)
//   * the count pass also adds up the keys of every 4096-key scan tile (LDS histogram per workgroup, one global
//     atomic per touched tile), so the scan is ONE launch of independent tiles: tile b adds the totals in front of it.
//   * the pair searches' count / scan / fill bodies ride in the first three launches as extra workgroups.
constexpr int kIdxScanTile = 4096;      // 1024 threads x 4 keys
constexpr int kIdxMaxTiles = 1024;      // n_keys1 <= 2^22

struct IndexParams {
    const int64_t *idx_local, *seg_off, *seg_base;
    int n_seg, n_rel;
    int64_t u_off[LGCN_MAX_REL], v_off[LGCN_MAX_REL];
    int64_t start[LGCN_MAX_REL + 1];          // prefix of the relations' edge counts
    int64_t n_nodes, n_keys1;                 // keys + 1
    unsigned long long *cnt;                  // [n_keys1 + kIdxMaxTiles], zero on entry and on completion:
                                              // per key (edges | slots handed out << 32), then the scan tiles' totals
    int32_t *uv;                              // [2 * total] global (u, v) of every edge; u = -1: dropped
    int32_t *rowptr, *col;
    int edge_blocks, scan_blocks, n_jobs;
    int job_blocks[5];                        // prefix of the pair jobs' row-block counts
    int32_t *clear_word;                      // optional word zeroed by the count pass
    PairsJobs jobs;
};

__device__ __forceinline__ int64_t to_global(const IndexParams &p, int64_t i) {
    int lo = 0, hi = p.n_seg;                 // last segment with seg_off[s] <= i (as k_graph_gather)
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (p.seg_off[mid] <= i) lo = mid; else hi = mid;
    }
    return p.idx_local[i] + p.seg_base[lo];
}

template <int PASS>     // 0: count, 1: fill
__global__ __launch_bounds__(256) void k_index_edges(const IndexParams p) {
    __shared__ int s_tile[kIdxMaxTiles];
    const int bid = blockIdx.x;
    if (PASS == 0 && bid == 0 && threadIdx.x == 0 && p.clear_word) *p.clear_word = 0;
    if (bid >= p.edge_blocks) {               // pair-search workgroups
        const int b = bid - p.edge_blocks;
        int jn = 0;
        while (jn + 1 < p.n_jobs && b >= p.job_blocks[jn + 1]) ++jn;
        const PairsJob &j = p.jobs.j[jn];
        pairs_rows_body<PASS>(j.agt, j.agt_off, j.ctx, j.ctx_off, j.n_scenes, j.n_agt, j.n_ctx, j.th, j.rp(), j.rp(), j.hi_base(),
                              j.wi_base(), j.hi, j.wi, j.cap, j.legacy, j.rowptr_q, b - p.job_blocks[jn]);
        return;
    }
    if (PASS == 0) {
        for (int i = threadIdx.x; i < p.scan_blocks; i += blockDim.x) s_tile[i] = 0;
        __syncthreads();
    }
    const int64_t total = p.start[p.n_rel];
    for (int64_t e = (int64_t)bid * blockDim.x + threadIdx.x; e < total; e += (int64_t)p.edge_blocks * blockDim.x) {
        int r = 0;
        while (r + 1 < p.n_rel && e >= p.start[r + 1]) ++r;
        if (PASS == 0) {
            const int64_t le = e - p.start[r];
            const int64_t u = to_global(p, p.u_off[r] + le), v = to_global(p, p.v_off[r] + le);
            const bool ok = u >= 0 && u < p.n_nodes && v >= 0 && v < p.n_nodes;     // never index out of bounds
            p.uv[2 * e] = ok ? (int32_t)u : -1;
            p.uv[2 * e + 1] = (int32_t)v;
            if (ok) {
                const int64_t k = csr_key(u, r, p.n_rel);
                atomicAdd(&p.cnt[k], 1ull);
                atomicAdd(&s_tile[k / kIdxScanTile], 1);
            }
        } else {
            const int u = p.uv[2 * e], v = p.uv[2 * e + 1];
            if (u < 0) continue;
            const int64_t k = csr_key(u, r, p.n_rel);
            const unsigned long long a = atomicAdd(&p.cnt[k], 1ull << 32);
            p.col[p.rowptr[k] + (int)(a >> 32)] = v;
        }
    }
    if (PASS == 0) {
        __syncthreads();
        for (int i = threadIdx.x; i < p.scan_blocks; i += blockDim.x)
            if (s_tile[i] != 0) atomicAdd(&p.cnt[p.n_keys1 + i], (unsigned long long)s_tile[i]);
    }
}

__global__ __launch_bounds__(1024) void k_index_scan(const IndexParams p) {
    __shared__ int lds[1024 / 64 + 1];
    const int bid = blockIdx.x;
    if (bid >= p.scan_blocks) {               // one workgroup per pair search: row scan + per-scene bases
        const PairsJob &j = p.jobs.j[bid - p.scan_blocks];
        pairs_scan_bases_body(j.rp(), j.n_agt, j.agt_off, j.ctx_off, j.n_scenes, j.legacy, j.cap, j.hi_base(), j.wi_base(),
                              j.n_pairs);
        if (j.n_agt == 0 && threadIdx.x == 0) j.rowptr_q[0] = 0;
        return;
    }
    // keys in front of this tile: the totals of the tiles in front of it (<= 1024: one per thread)
    const int part = (int)threadIdx.x < bid ? (int)p.cnt[p.n_keys1 + threadIdx.x] : 0;
    int before;
    block_exclusive_scan<1024>(part, &before, lds);
    const int64_t b = (int64_t)bid * kIdxScanTile + 4 * (int64_t)threadIdx.x;
    int v[4], sum = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        v[k] = b + k < p.n_keys1 ? (int)(p.cnt[b + k] & 0xffffffffull) : 0;
        sum += v[k];
    }
    int tile_total;
    int off = block_exclusive_scan<1024>(sum, &tile_total, lds) + before;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (b + k < p.n_keys1) p.rowptr[b + k] = off;
        off += v[k];
    }
}

// canonical order inside each key (as k_csr_sort_rows) + the counters back to zero
__global__ __launch_bounds__(256) void k_index_sort(const IndexParams p) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < kIdxMaxTiles) p.cnt[p.n_keys1 + k] = 0ull;
    if (k >= p.n_keys1) return;
    const unsigned long long w = p.cnt[k];
    if (w == 0ull) return;
    p.cnt[k] = 0ull;
    const int n = (int)(w & 0xffffffffull);
    int32_t *c = p.col + p.rowptr[k];
    for (int i = 1; i < n; ++i) {
        const int x = c[i];
        int q = i - 1;
        while (q >= 0 && c[q] > x) { c[q + 1] = c[q]; --q; }
        c[q + 1] = x;
    }
}

// -------------------------------------------- graph construction (row f3) -----
// dilated_nbrs (reference data.py:520-534): the scale-i relation is the boolean power A^(2^i) of the scale-0
// adjacency, formed by repeated squaring (mat = mat * mat).  One squaring of a CSR matrix with sorted or unsorted
// rows (duplicates allowed: they only repeat candidates):
//   bound  : cand_ptr[u] = exclusive scan of sum_{v in A[u]} |A[v]|
//   expand : row u's candidates {w : w in A[v], v in A[u]} -> sorted, duplicates removed in place, count kept
//   compact: rows packed to the scanned counts (+ the COO row index of every entry)
// One thread per row: lane graphs branch rarely, a row of A^32 has a handful of entries.
__global__ __launch_bounds__(256) void k_sq_bound(const int32_t *rowptr, const int32_t *col, int64_t n, int32_t *ub) {
    const int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (u > n) return;
    int s = 0;
    if (u < n)
        for (int e = rowptr[u]; e < rowptr[u + 1]; ++e) {
            const int v = col[e];
            if (v >= 0 && v < n) s += rowptr[v + 1] - rowptr[v];
        }
    ub[u] = s;        // ub[n] = 0: the scan leaves the total there
}

__global__ __launch_bounds__(256) void k_sq_expand(const int32_t *rowptr, const int32_t *col, int64_t n,
                                                   const int32_t *cand_ptr, int32_t *cand, int32_t *cnt) {
    const int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (u > n) return;
    if (u == n) { cnt[n] = 0; return; }
    int32_t *c = cand + cand_ptr[u];
    int k = 0;
    for (int e = rowptr[u]; e < rowptr[u + 1]; ++e) {
        const int v = col[e];
        if (v < 0 || v >= n) continue;
        for (int f = rowptr[v]; f < rowptr[v + 1]; ++f) {      // insertion into the sorted, duplicate-free prefix
            const int w = col[f];
            int q = k - 1;
            while (q >= 0 && c[q] > w) --q;
            if (q >= 0 && c[q] == w) continue;
            for (int r = k; r > q + 1; --r) c[r] = c[r - 1];
            c[q + 1] = w;
            ++k;
        }
    }
    cnt[u] = k;
}

__global__ __launch_bounds__(256) void k_sq_compact(const int32_t *cand_ptr, const int32_t *cand, const int32_t *out_rowptr,
                                                    int64_t n, int32_t *out_col, int32_t *out_row) {
    const int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= n) return;
    const int b = out_rowptr[u], k = out_rowptr[u + 1] - b;
    const int32_t *c = cand + cand_ptr[u];
    for (int i = 0; i < k; ++i) {
        out_col[b + i] = c[i];
        if (out_row) out_row[b + i] = (int32_t)u;
    }
}

// Left / right node adjacency (reference preprocess_data.py:287-392, cross_angle = None).
// Lane-level mask (:318-327 / :356-360): mat = (S pre + S suc + S) > 0.5 with S, pre, suc the 0/1 matrices of the side
// pairs and the predecessor / successor lane pairs: mat[a][b] = S[a][b] or exists c: S[a][c] and (pre[c][b] or suc[c][b]).
__global__ __launch_bounds__(256) void k_lane_mask(const int64_t *side, int64_t n_side, const int64_t *pre, int64_t n_pre,
                                                   const int64_t *suc, int64_t n_suc, int num_lanes, uint8_t *mat) {
    const int64_t per = n_pre + n_suc + 1;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_side * per) return;
    const int64_t i = t / per, j = t % per;
    const int64_t a = side[2 * i], c = side[2 * i + 1];
    if (a < 0 || a >= num_lanes || c < 0 || c >= num_lanes) return;
    int64_t b = c;
    if (j > 0) {
        const int64_t *pp = j - 1 < n_pre ? pre + 2 * (j - 1) : suc + 2 * (j - 1 - n_pre);
        if (pp[0] != c) return;
        b = pp[1];
        if (b < 0 || b >= num_lanes) return;
    }
    mat[a * num_lanes + b] = 1;
}

// One wave per node h (:329-347): over the nodes w whose lane is allowed for h's lane, the nearest centre (fp32
// sub / mul / add / sqrt as ATen evaluates sqrt(((a - b) ** 2).sum(2)); the FIRST w among equals, as a CPU min(1)
// returns); kept when the distance is < cross_dist and the two segments' headings differ by < pi / 4.
__global__ __launch_bounds__(256) void k_cross_edges(const float2 *ctrs, const float2 *feats, const int64_t *lane_idcs,
                                                     int n, const uint8_t *mat, int num_lanes, float cross_dist,
                                                     int32_t *partner) {
    const int lane = threadIdx.x & 63;
    const int h = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (h >= n) return;
    const float2 a = ctrs[h];
    const int64_t la = lane_idcs[h];
    const bool la_ok = la >= 0 && la < num_lanes;
    float best = 3.0e38f;
    int bw = 0x7fffffff;
    for (int w0 = 0; w0 < n; w0 += 64) {
        const int w = w0 + lane;
        if (w < n) {
            const int64_t lb = lane_idcs[w];
            const bool ok = la_ok && lb >= 0 && lb < num_lanes && mat[la * num_lanes + lb] != 0;
            float d = 1e6f;                                     // :332 masked entries
            if (ok) {
                const float2 c = ctrs[w];
                const float dx = __fsub_rn(a.x, c.x), dy = __fsub_rn(a.y, c.y);
                d = __fsqrt_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)));
            }
            if (d < best) { best = d; bw = w; }                 // ascending w per lane: the first of equals stays
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const float ob = __shfl_xor(best, off, 64);
        const int ow = __shfl_xor(bw, off, 64);
        if (ob < best || (ob == best && ow < bw)) { best = ob; bw = ow; }
    }
    if (lane == 0) {
        int out = -1;
        if (best < cross_dist && bw < n) {
            const float2 f1 = feats[h], f2 = feats[bw];
            const float t1 = atan2f(f1.y, f1.x), t2 = atan2f(f2.y, f2.x);
            float dt = fabsf(__fsub_rn(t1, t2));
            if (dt > 3.14159274101257324f) dt = fabsf(__fsub_rn(dt, 6.28318548202514648f));      // float32(pi), float32(2 pi)
            if (dt < 0.785398185253143311f) out = bw;                                            // float32(pi / 4)
        }
        partner[h] = out;
    }
}

// rowptr and cursor of the CSR plan zeroed in one launch (two memset nodes cost a launch boundary each)
__global__ __launch_bounds__(256) void k_zero2(int32_t *a, int32_t *b, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        a[i] = 0;
        b[i] = 0;
    }
}

__global__ __launch_bounds__(256) void k_widen(const int32_t *in, const int32_t *n_dev, int64_t cap, int64_t *out) {
    int64_t n = *n_dev;
    if (n < 0 || n > cap) n = cap;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = in[i];
}

static inline unsigned grid_for(int64_t n, int threads, int64_t max_blocks = 4096) {
    int64_t b = (n + threads - 1) / threads;
    if (b < 1) b = 1;
    if (b > max_blocks) b = max_blocks;
    return (unsigned)b;
}

}  // namespace lgcn

using namespace lgcn;

extern "C" {

int lgcn_version(void) { return LGCN_VERSION; }

const char *lgcn_strerror(int code) {
    switch (code) {
        case LGCN_OK: return "ok";
        case LGCN_EINVAL: return "invalid argument (null pointer, negative size or bad flag)";
        case LGCN_ESHAPE: return "unsupported shape";
        case LGCN_EALIGN: return "pointer not 16-byte aligned";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown lgcn error";
    }
}

int lgcn_graph_gather(const int64_t *in, int64_t n_elem, const int64_t *seg_off, const int64_t *seg_base,
                      int n_seg, int64_t *out64, int32_t *out32, void *stream) {
    if (n_elem < 0 || n_seg < 0) return LGCN_EINVAL;
    if (n_elem == 0) return LGCN_OK;
    LGCN_CHECK_PTR(in); LGCN_CHECK_PTR(seg_off); LGCN_CHECK_PTR(seg_base);
    if (n_seg < 1 || (!out64 && !out32)) return LGCN_EINVAL;
    hipLaunchKernelGGL(k_graph_gather, dim3(grid_for(n_elem, 256)), dim3(256), 0, (hipStream_t)stream,
                       in, n_elem, seg_off, seg_base, n_seg, out64, out32);
    return launch_status();
}

int64_t lgcn_csr_rowptr_elems(int64_t n_nodes, int n_rel) {
    if (n_nodes < 0 || n_rel < 1 || n_rel > LGCN_MAX_REL) return LGCN_EINVAL;
    return ((n_nodes + 15) / 16) * n_rel * 16 + 1;
}

int64_t lgcn_csr_ws_elems(int64_t n_nodes, int n_rel) {
    const int64_t k = lgcn_csr_rowptr_elems(n_nodes, n_rel);
    if (k < 0) return k;
    return k + scan_ws_elems(k);  // cursor + scan block sums
}

int lgcn_csr_build(const int64_t *const *u_host, const int64_t *const *v_host, const int64_t *n_edges_host,
                   int n_rel, int64_t n_nodes, int32_t *rowptr, int32_t *col, int32_t *ws, void *stream) {
    if (n_rel < 1 || n_rel > LGCN_MAX_REL || n_nodes < 0) return LGCN_EINVAL;
    LGCN_CHECK_PTR(u_host); LGCN_CHECK_PTR(v_host); LGCN_CHECK_PTR(n_edges_host);
    LGCN_CHECK_PTR(rowptr); LGCN_CHECK_PTR(ws);
    hipStream_t st = (hipStream_t)stream;
    CooTable t;
    t.n_rel = n_rel;
    t.n_nodes = n_nodes;
    t.start[0] = 0;
    for (int r = 0; r < n_rel; ++r) {
        if (n_edges_host[r] < 0) return LGCN_EINVAL;
        if (n_edges_host[r] > 0 && (!u_host[r] || !v_host[r])) return LGCN_EINVAL;
        t.u[r] = u_host[r];
        t.v[r] = v_host[r];
        t.start[r + 1] = t.start[r] + n_edges_host[r];
    }
    for (int r = n_rel; r < LGCN_MAX_REL; ++r) { t.u[r] = nullptr; t.v[r] = nullptr; t.start[r + 1] = t.start[n_rel]; }
    const int64_t total = t.start[n_rel];
    if (total > 0x7fffffff || n_nodes > 0x7fffffff) return LGCN_ESHAPE;
    if (total > 0) LGCN_CHECK_PTR(col);
    const int64_t nk1 = lgcn_csr_rowptr_elems(n_nodes, n_rel);  // keys + 1
    int32_t *cursor = ws;
    int32_t *sums = ws + nk1;
    hipLaunchKernelGGL(k_zero2, dim3(grid_for(nk1, 256, 1024)), dim3(256), 0, st, rowptr, cursor, nk1);
    if (total > 0) {
        hipLaunchKernelGGL((k_csr_edges<0>), dim3(grid_for(total, 256)), dim3(256), 0, st, t, rowptr, nullptr, nullptr);
    }
    int rc = exclusive_scan(rowptr, rowptr, nk1, sums, st);
    if (rc != LGCN_OK) return rc;
    if (total > 0) {
        hipLaunchKernelGGL((k_csr_edges<1>), dim3(grid_for(total, 256)), dim3(256), 0, st, t, cursor, rowptr, col);
        hipLaunchKernelGGL(k_csr_sort_rows, dim3((unsigned)((nk1 - 1 + 255) / 256)), dim3(256), 0, st,
                           rowptr, col, nk1 - 1);
    }
    return launch_status();
}

int64_t lgcn_pairs_ws_elems(int64_t n_agt, int n_scenes) {
    if (n_agt < 0 || n_scenes < 0) return LGCN_EINVAL;
    // rowptr_true [T+1] + hi_base [B+1] + wi_base [B]
    return (n_agt + 1) + 2 * (int64_t)n_scenes + 1;
}

static int pairs_job_check(const lgcn_pairs_job_t &q) {
    if (q.n_scenes < 1 || q.n_agt < 0 || q.n_ctx < 0 || q.cap < 0) return LGCN_EINVAL;
    if (q.n_agt > 0x7ffffff0 || q.n_ctx > 0x7ffffff0 || q.cap > 0x7ffffff0) return LGCN_ESHAPE;
    LGCN_CHECK_PTR(q.agt_off); LGCN_CHECK_PTR(q.ctx_off); LGCN_CHECK_PTR(q.n_pairs);
    LGCN_CHECK_PTR(q.rowptr); LGCN_CHECK_PTR(q.ws);
    if (q.n_agt > 0) LGCN_CHECK_PTR(q.agt_ctrs);
    if (q.n_ctx > 0) LGCN_CHECK_PTR(q.ctx_ctrs);
    if (q.cap > 0) { LGCN_CHECK_PTR(q.hi); LGCN_CHECK_PTR(q.wi); }
    return LGCN_OK;
}

int lgcn_pairs_build_multi(const lgcn_pairs_job_t *jobs, int n_jobs, void *stream) {
    if (n_jobs < 1 || n_jobs > 4) return LGCN_EINVAL;
    LGCN_CHECK_PTR(jobs);
    hipStream_t st = (hipStream_t)stream;
    PairsJobs dj;
    int64_t max_rows = 0;
    for (int k = 0; k < n_jobs; ++k) {
        const lgcn_pairs_job_t &q = jobs[k];
        const int rc = pairs_job_check(q);
        if (rc != LGCN_OK) return rc;
        PairsJob &d = dj.j[k];
        d.agt = (const float2 *)q.agt_ctrs; d.agt_off = q.agt_off;
        d.ctx = (const float2 *)q.ctx_ctrs; d.ctx_off = q.ctx_off;
        d.n_scenes = q.n_scenes; d.n_agt = (int)q.n_agt; d.n_ctx = (int)q.n_ctx; d.legacy = q.legacy_offsets; d.th = q.dist_th;
        d.hi = q.hi; d.wi = q.wi; d.cap = q.cap; d.n_pairs = q.n_pairs; d.rowptr_q = q.rowptr; d.ws = q.ws;
        if (q.n_agt > max_rows) max_rows = q.n_agt;
    }
    for (int k = n_jobs; k < 4; ++k) dj.j[k] = dj.j[0];
    // three launches for all jobs: count per target row, scan + per-scene bases (one block per job), fill
    // (+ segment table).  A job without rows still gets rowptr[0] = P = 0 from the scan block.
    const unsigned row_blocks = (unsigned)((max_rows + 3) / 4);
    if (max_rows > 0) hipLaunchKernelGGL((k_pairs_rows<0>), dim3(row_blocks, n_jobs), dim3(256), 0, st, dj);
    hipLaunchKernelGGL(k_pairs_scan_bases, dim3(n_jobs), dim3(1024), 0, st, dj);
    if (max_rows > 0) hipLaunchKernelGGL((k_pairs_rows<1>), dim3(row_blocks, n_jobs), dim3(256), 0, st, dj);
    return launch_status();
}

int lgcn_pairs_build(const float *agt_ctrs, const int32_t *agt_off, const float *ctx_ctrs,
                     const int32_t *ctx_off, int n_scenes, int64_t n_agt, int64_t n_ctx, float dist_th,
                     int legacy_offsets, int32_t *hi, int32_t *wi, int64_t cap, int32_t *n_pairs,
                     int32_t *rowptr, int32_t *ws, void *stream) {
    lgcn_pairs_job_t q;
    q.agt_ctrs = agt_ctrs; q.agt_off = agt_off; q.ctx_ctrs = ctx_ctrs; q.ctx_off = ctx_off;
    q.n_scenes = n_scenes; q.legacy_offsets = legacy_offsets; q.n_agt = n_agt; q.n_ctx = n_ctx;
    q.dist_th = dist_th; q.pad_ = 0; q.hi = hi; q.wi = wi; q.cap = cap; q.n_pairs = n_pairs; q.rowptr = rowptr; q.ws = ws;
    return lgcn_pairs_build_multi(&q, 1, stream);
}

int64_t lgcn_index_uv_elems(int64_t n_edges) { return n_edges < 0 ? (int64_t)LGCN_EINVAL : 2 * n_edges; }

int64_t lgcn_index_cnt_words(int64_t n_nodes, int n_rel) {
    const int64_t k = lgcn_csr_rowptr_elems(n_nodes, n_rel);
    return k < 0 ? k : k + kIdxMaxTiles;
}

int lgcn_index_build(const lgcn_index_t *ph, void *stream) {
    LGCN_CHECK_PTR(ph);
    const lgcn_index_t &q = *ph;
    if (q.n_rel < 1 || q.n_rel > LGCN_MAX_REL || q.n_nodes < 0 || q.n_seg < 1 || q.n_jobs < 0 || q.n_jobs > 4 || q.n_elem < 0)
        return LGCN_EINVAL;
    LGCN_CHECK_PTR(q.rowptr); LGCN_CHECK_PTR(q.cnt);
    if (((uintptr_t)q.cnt & 7u) != 0) return LGCN_EALIGN;
    if (q.n_jobs > 0) LGCN_CHECK_PTR(q.jobs);
    IndexParams p;
    p.idx_local = q.idx_local; p.seg_off = q.seg_off; p.seg_base = q.seg_base; p.n_seg = q.n_seg; p.n_rel = q.n_rel;
    p.start[0] = 0;
    for (int r = 0; r < q.n_rel; ++r) {
        if (q.n_edges[r] < 0 || q.u_off[r] < 0 || q.v_off[r] < 0 || q.u_off[r] + q.n_edges[r] > q.n_elem ||
            q.v_off[r] + q.n_edges[r] > q.n_elem)
            return LGCN_EINVAL;
        p.u_off[r] = q.u_off[r]; p.v_off[r] = q.v_off[r];
        p.start[r + 1] = p.start[r] + q.n_edges[r];
    }
    for (int r = q.n_rel; r < LGCN_MAX_REL; ++r) { p.u_off[r] = p.v_off[r] = 0; p.start[r + 1] = p.start[q.n_rel]; }
    const int64_t total = p.start[q.n_rel];
    if (total > 0x7fffffff || q.n_nodes > 0x7fffffff) return LGCN_ESHAPE;
    const int64_t nk1 = lgcn_csr_rowptr_elems(q.n_nodes, q.n_rel);
    if (nk1 > (int64_t)kIdxScanTile * kIdxMaxTiles) return LGCN_ESHAPE;      // one scan launch: <= 1024 tiles of 4096 keys
    if (total > 0) {
        LGCN_CHECK_PTR(q.idx_local); LGCN_CHECK_PTR(q.seg_off); LGCN_CHECK_PTR(q.seg_base);
        LGCN_CHECK_PTR(q.col); LGCN_CHECK_PTR(q.uv);
    }
    p.n_nodes = q.n_nodes; p.n_keys1 = nk1;
    p.cnt = reinterpret_cast<unsigned long long *>(q.cnt); p.uv = q.uv; p.rowptr = q.rowptr; p.col = q.col;
    p.n_jobs = q.n_jobs;
    p.clear_word = q.clear_word;
    p.job_blocks[0] = 0;
    for (int k = 0; k < 4; ++k) {
        if (k < q.n_jobs) {
            const lgcn_pairs_job_t &jq = q.jobs[k];
            const int rc = pairs_job_check(jq);
            if (rc != LGCN_OK) return rc;
            PairsJob &d = p.jobs.j[k];
            d.agt = (const float2 *)jq.agt_ctrs; d.agt_off = jq.agt_off;
            d.ctx = (const float2 *)jq.ctx_ctrs; d.ctx_off = jq.ctx_off;
            d.n_scenes = jq.n_scenes; d.n_agt = (int)jq.n_agt; d.n_ctx = (int)jq.n_ctx; d.legacy = jq.legacy_offsets; d.th = jq.dist_th;
            d.hi = jq.hi; d.wi = jq.wi; d.cap = jq.cap; d.n_pairs = jq.n_pairs; d.rowptr_q = jq.rowptr; d.ws = jq.ws;
            p.job_blocks[k + 1] = p.job_blocks[k] + (int)((jq.n_agt + 3) / 4);
        } else {
            if (k > 0) p.jobs.j[k] = p.jobs.j[0];
            p.job_blocks[k + 1] = p.job_blocks[k];
        }
    }
    if (q.n_jobs == 0) {
        PairsJob z{};
        for (int k = 0; k < 4; ++k) p.jobs.j[k] = z;
    }
    p.edge_blocks = total > 0 ? (int)grid_for(total, 256, 1024) : 0;
    p.scan_blocks = (int)((nk1 + kIdxScanTile - 1) / kIdxScanTile);
    hipStream_t st = (hipStream_t)stream;
    const unsigned g13 = (unsigned)(p.edge_blocks + p.job_blocks[4]);
    if (g13 > 0) hipLaunchKernelGGL((k_index_edges<0>), dim3(g13), dim3(256), 0, st, p);
    else if (q.clear_word) { hipError_t e = hipMemsetAsync(q.clear_word, 0, 4, st); if (e != hipSuccess) return (int)e; }
    hipLaunchKernelGGL(k_index_scan, dim3((unsigned)(p.scan_blocks + q.n_jobs)), dim3(1024), 0, st, p);
    if (g13 > 0) hipLaunchKernelGGL((k_index_edges<1>), dim3(g13), dim3(256), 0, st, p);
    if (total > 0) hipLaunchKernelGGL(k_index_sort, dim3((unsigned)((nk1 + 255) / 256)), dim3(256), 0, st, p);
    return launch_status();
}

int lgcn_bool_square_bound(const int32_t *rowptr, const int32_t *col, int64_t n, int32_t *cand_ptr, int32_t *ws,
                            void *stream) {
    if (n < 0) return LGCN_EINVAL;
    if (n > 0x7ffffff0) return LGCN_ESHAPE;
    LGCN_CHECK_PTR(rowptr); LGCN_CHECK_PTR(cand_ptr); LGCN_CHECK_PTR(ws);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_sq_bound, dim3(grid_for(n + 1, 256, 1 << 22)), dim3(256), 0, st, rowptr, col, n, cand_ptr);
    return exclusive_scan(cand_ptr, cand_ptr, n + 1, ws, st);
}

int lgcn_bool_square(const int32_t *rowptr, const int32_t *col, int64_t n, const int32_t *cand_ptr, int32_t *cand,
                     int32_t *out_rowptr, int32_t *ws, void *stream) {
    if (n < 0) return LGCN_EINVAL;
    if (n > 0x7ffffff0) return LGCN_ESHAPE;
    LGCN_CHECK_PTR(rowptr); LGCN_CHECK_PTR(cand_ptr); LGCN_CHECK_PTR(out_rowptr); LGCN_CHECK_PTR(ws);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_sq_expand, dim3(grid_for(n + 1, 256, 1 << 22)), dim3(256), 0, st, rowptr, col, n, cand_ptr, cand,
                       out_rowptr);
    return exclusive_scan(out_rowptr, out_rowptr, n + 1, ws, st);
}

int lgcn_bool_square_compact(const int32_t *cand_ptr, const int32_t *cand, const int32_t *out_rowptr, int64_t n,
                             int32_t *out_col, int32_t *out_row, void *stream) {
    if (n < 0) return LGCN_EINVAL;
    if (n == 0) return LGCN_OK;
    LGCN_CHECK_PTR(cand_ptr); LGCN_CHECK_PTR(out_rowptr);
    hipLaunchKernelGGL(k_sq_compact, dim3(grid_for(n, 256, 1 << 22)), dim3(256), 0, (hipStream_t)stream, cand_ptr, cand,
                       out_rowptr, n, out_col, out_row);
    return launch_status();
}

int64_t lgcn_scan_ws_elems(int64_t n) { return n < 0 ? (int64_t)LGCN_EINVAL : scan_ws_elems(n); }

int lgcn_cross_edges(const float *ctrs, const float *feats, const int64_t *lane_idcs, int64_t n_nodes, int num_lanes,
                     const int64_t *side_pairs, int64_t n_side, const int64_t *pre_pairs, int64_t n_pre,
                     const int64_t *suc_pairs, int64_t n_suc, float cross_dist, uint8_t *mat, int32_t *partner,
                     void *stream) {
    if (n_nodes < 0 || num_lanes < 0 || n_side < 0 || n_pre < 0 || n_suc < 0) return LGCN_EINVAL;
    if (n_nodes > 0x7ffffff0 || (int64_t)num_lanes * num_lanes > 0x7ffffff0) return LGCN_ESHAPE;
    if (n_nodes == 0) return LGCN_OK;
    LGCN_CHECK_PTR(ctrs); LGCN_CHECK_PTR(feats); LGCN_CHECK_PTR(lane_idcs); LGCN_CHECK_PTR(partner);
    if (num_lanes > 0) LGCN_CHECK_PTR(mat);
    if (n_side > 0) LGCN_CHECK_PTR(side_pairs);
    if (n_pre > 0) LGCN_CHECK_PTR(pre_pairs);
    if (n_suc > 0) LGCN_CHECK_PTR(suc_pairs);
    hipStream_t st = (hipStream_t)stream;
    if (num_lanes > 0) {
        hipError_t e = hipMemsetAsync(mat, 0, (size_t)num_lanes * num_lanes, st);
        if (e != hipSuccess) return (int)e;
    }
    const int64_t work = n_side * (n_pre + n_suc + 1);
    if (work > 0)
        hipLaunchKernelGGL(k_lane_mask, dim3(grid_for(work, 256, 1 << 22)), dim3(256), 0, st, side_pairs, n_side, pre_pairs,
                           n_pre, suc_pairs, n_suc, num_lanes, mat);
    hipLaunchKernelGGL(k_cross_edges, dim3((unsigned)((n_nodes + 3) / 4)), dim3(256), 0, st, (const float2 *)ctrs,
                       (const float2 *)feats, lane_idcs, (int)n_nodes, mat, num_lanes, cross_dist, partner);
    return launch_status();
}

int lgcn_widen_i32(const int32_t *in, const int32_t *n_dev, int64_t cap, int64_t *out, void *stream) {
    if (cap < 0) return LGCN_EINVAL;
    if (cap == 0) return LGCN_OK;
    LGCN_CHECK_PTR(in); LGCN_CHECK_PTR(n_dev); LGCN_CHECK_PTR(out);
    hipLaunchKernelGGL(k_widen, dim3(grid_for(cap, 256, 1024)), dim3(256), 0, (hipStream_t)stream, in, n_dev, cap, out);
    return launch_status();
}

}  // extern "C"
