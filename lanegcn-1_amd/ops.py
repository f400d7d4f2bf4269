"""Tensor-level wrappers over the C ABI (include/lgcn.h).

Every function takes CUDA (ROCm) tensors, enqueues on the current torch stream and returns
without synchronising.  CPU tensors are rejected: the product path has no CPU fallback.
"""
import ctypes as C
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib as L

import os

C_FEAT = 128
EPS = 1e-5

# How the 128-d contractions run on the matrix cores (include/lgcn.h, LGCN_MMA_*):
#   "f32"    exact fp32 fma chain (v_mfma_f32_32x32x2_f32)
#   "bf16x3" 3-way bf16 split, 6 products, fp32 accumulate: fp32-grade, 2.67x the f32 MFMA rate
#   "f16x2"  2-way fp16 split, 3 products, fp32 accumulate: fp32-grade (operands < 65504), 5.3x the f32 rate
#   "bf16"   single bf16 product (BASELINE config "bf16"; not within 1e-4)
_mma = L.MMA_NAMES[os.environ.get("LGCN_MMA", "f16x2")]


def set_mma(name: str):
    """Select the matrix-core mode for subsequent launches ("f32" | "bf16x3" | "f16x2" | "bf16")."""
    global _mma
    _mma = L.MMA_NAMES[name]


def get_mma() -> str:
    return {v: k for k, v in L.MMA_NAMES.items()}[_mma]


class backward_mma:
    """Matrix-core mode for the row-GEMMs of a backward pass: gradients routinely sit below fp16's normal
    range (6e-5), where the 2-way fp16 split loses its second plane, so f16x2 forwards run their backward
    GEMMs on the range-safe 3-way bf16 split (same fp32-grade accuracy, fp32 exponent range)."""

    def __enter__(self):
        global _mma
        self.prev = _mma
        if _mma == L.MMA_F16X2:
            _mma = L.MMA_BF16X3
        return self

    def __exit__(self, *a):
        global _mma
        _mma = self.prev


class mma_scope:
    """``with ops.mma_scope("bf16x3"):`` -- matrix-core mode for the launches inside the block."""

    def __init__(self, name: str):
        self.name = name

    def __enter__(self):
        global _mma
        self.prev = _mma
        _mma = L.MMA_NAMES[self.name]
        return self

    def __exit__(self, *a):
        global _mma
        _mma = self.prev


# What happens when a forward in the range-restricted default mode (f16x2: operands below 65504) comes back with
# non-finite features: "reroute" (default) runs it again in bf16x3 (fp32's exponent range, same fp32-grade
# accuracy), "raise" raises LgcnError, "off" returns it as it is.
_guard = os.environ.get("LGCN_GUARD", "reroute")


def set_guard(policy: str):
    global _guard
    if policy not in ("reroute", "raise", "off"):
        raise L.LgcnError("guard policy must be 'reroute', 'raise' or 'off'")
    _guard = policy


def get_guard() -> str:
    return _guard


def check_finite(flag: torch.Tensor, a: torch.Tensor, b: Optional[torch.Tensor] = None, bit: int = 1):
    """flag[0] |= bit when a (or b) holds a NaN / inf (lgcn_check_finite); enqueued, no synchronisation."""
    lib = L.load()
    a = _dev(a, torch.float32, "a")
    b = None if b is None else _dev(b, torch.float32, "b")
    L.check(lib.lgcn_check_finite(_ptr(a), a.numel(), _ptr(b), 0 if b is None else b.numel(), _ptr(flag), bit, _stream()),
            "lgcn_check_finite")


def guarded(run, tensors_of=lambda out: (out,)):
    """run() in the current matrix mode; in f16x2 the result's features are checked on the device (one 4-byte
    device->host read) and a forward that overflowed fp16's range is run again in bf16x3 -- or raises, by policy.
    Non-finite values that survive the re-run were in the inputs: returned as they are, like the reference does."""
    out = run()
    if _guard == "off" or _mma != L.MMA_F16X2:
        return out
    ts = [t for t in tensors_of(out) if t is not None and t.numel() > 0]
    if not ts:
        return out
    flag = torch.zeros(1, dtype=torch.int32, device=ts[0].device)
    for i in range(0, len(ts), 2):
        check_finite(flag, ts[i], ts[i + 1] if i + 1 < len(ts) else None)
    if int(flag.item()) == 0:
        return out
    if _guard == "raise":
        raise L.LgcnError("non-finite features in f16x2 mode: an operand left fp16's range (|x| >= 65504); "
                          "use ops.set_mma('bf16x3') or ops.set_guard('reroute')")
    with mma_scope("bf16x3"):
        return run()


def _stream():
    # the raw hipStream_t of torch's current stream (torch.cuda.current_stream() costs ~9 us of Python per call)
    return C.c_void_p(torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))


def module_params(module) -> tuple:
    """tuple(module.parameters()), cached on the module (walking the module tree costs more than a launch)."""
    ps = module.__dict__.get("_lgcn_params")
    if ps is None:
        ps = tuple(module.parameters())
        module.__dict__["_lgcn_params"] = ps
    return ps


def _dev(t: torch.Tensor, dtype=None, name="tensor"):
    if not t.is_cuda:
        raise L.LgcnError("%s must be a CUDA tensor (the HIP hot path has no CPU fallback)" % name)
    if dtype is not None and t.dtype != dtype:
        raise L.LgcnError("%s must be %s, got %s" % (name, dtype, t.dtype))
    return t if t.is_contiguous() else t.contiguous()


def _ptr(t: Optional[torch.Tensor]):
    return C.c_void_p(0 if t is None else t.data_ptr())


def wants_grad(*tensors) -> bool:
    """True when autograd must record this call (then the differentiable path of autograd.py runs)."""
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)


# ------------------------------------------------------------------ integer path
def graph_gather_indices(flat_in: torch.Tensor, seg_off: torch.Tensor, seg_base: torch.Tensor, want32=False):
    """out[e] = in[e] + base[seg(e)] (lanegcn.py:191-208).  Returns (out64, out32|None)."""
    lib = L.load()
    flat_in = _dev(flat_in, torch.int64, "flat_in")
    seg_off = _dev(seg_off, torch.int64, "seg_off")
    seg_base = _dev(seg_base, torch.int64, "seg_base")
    n = flat_in.numel()
    out64 = torch.empty_like(flat_in)
    out32 = torch.empty(n, dtype=torch.int32, device=flat_in.device) if want32 else None
    L.check(lib.lgcn_graph_gather(_ptr(flat_in), n, _ptr(seg_off), _ptr(seg_base), seg_base.numel(),
                                  _ptr(out64), _ptr(out32), _stream()), "lgcn_graph_gather")
    return out64, out32


@dataclass
class LanePlan:
    """Tile-major CSR-by-destination of the lane relations (see include/lgcn.h)."""
    rowptr: torch.Tensor
    col: torch.Tensor
    n_rel: int
    n_nodes: int
    n_edges: List[int]


def csr_build(u_list: Sequence[torch.Tensor], v_list: Sequence[torch.Tensor], n_nodes: int) -> LanePlan:
    lib = L.load()
    n_rel = len(u_list)
    if n_rel < 1 or n_rel > L.MAX_REL or len(v_list) != n_rel:
        raise L.LgcnError("csr_build: need 1..%d relations" % L.MAX_REL)
    us = [_dev(u, torch.int64, "u") for u in u_list]
    vs = [_dev(v, torch.int64, "v") for v in v_list]
    ne = [int(u.numel()) for u in us]
    for u, v in zip(us, vs):
        if u.numel() != v.numel():
            raise L.LgcnError("csr_build: u and v differ in length")
    dev = us[0].device
    nk1 = lib.lgcn_csr_rowptr_elems(n_nodes, n_rel)
    rowptr = torch.empty(nk1, dtype=torch.int32, device=dev)
    col = torch.empty(max(sum(ne), 1), dtype=torch.int32, device=dev)
    ws = torch.empty(lib.lgcn_csr_ws_elems(n_nodes, n_rel), dtype=torch.int32, device=dev)
    up = (C.c_void_p * n_rel)(*[u.data_ptr() if u.numel() else 0 for u in us])
    vp = (C.c_void_p * n_rel)(*[v.data_ptr() if v.numel() else 0 for v in vs])
    nn = (C.c_int64 * n_rel)(*ne)
    L.check(lib.lgcn_csr_build(up, vp, nn, n_rel, n_nodes, _ptr(rowptr), _ptr(col), _ptr(ws), _stream()),
            "lgcn_csr_build")
    return LanePlan(rowptr, col, n_rel, n_nodes, ne)


@dataclass
class PairSet:
    """Result of the Att pair search (lanegcn.py:672-689) kept on device."""
    hi: torch.Tensor          # [cap] int32, first P valid
    wi: torch.Tensor          # [cap] int32
    n_pairs: torch.Tensor     # [1] int32 (device)
    rowptr: torch.Tensor      # [T+1] int32: segments of index_add_(0, hi, .)
    cap: int
    n_agt: int
    agt_ctrs: torch.Tensor    # [T,2] concatenated centres
    ctx_ctrs: torch.Tensor    # [S,2]
    _p_host: Optional[int] = None

    def count(self) -> int:
        """P on the host (one device->host read, cached)."""
        if self._p_host is None:
            self._p_host = int(self.n_pairs.item())
            if self._p_host < 0:
                raise L.LgcnError("pair capacity %d exceeded (P = %d)" % (self.cap, -self._p_host))
        return self._p_host

    def csr_by_wi(self, n_ctx: int):
        """Plain CSR of the pairs by context row (rowptr [S+1], col = pair ids, int32): the transpose of the
        gather V[wi] needed by the backward.  Index plumbing on ATen integer ops, cached."""
        if getattr(self, "_wcsr", None) is None:
            P = self.count()
            wi = self.wi[:P].long()
            order = torch.argsort(wi, stable=True)
            # counts without torch.bincount (it reads the maximum back to the host: ~0.7 ms of sync per call)
            counts = torch.zeros(n_ctx, dtype=torch.int64, device=wi.device).index_add_(0, wi, torch.ones_like(wi))
            rowptr = torch.zeros(n_ctx + 1, dtype=torch.int32, device=wi.device)
            rowptr[1:] = torch.cumsum(counts, 0).to(torch.int32)
            self._wcsr = (rowptr, order.to(torch.int32).contiguous())
        return self._wcsr

    def hi_wi_long(self):
        """(hi, wi) as the reference's LongTensors [P] (test/debug accessor; syncs once)."""
        lib = L.load()
        P = self.count()
        h = torch.empty(self.cap, dtype=torch.int64, device=self.hi.device)
        w = torch.empty_like(h)
        L.check(lib.lgcn_widen_i32(_ptr(self.hi), _ptr(self.n_pairs), self.cap, _ptr(h), _stream()), "lgcn_widen_i32")
        L.check(lib.lgcn_widen_i32(_ptr(self.wi), _ptr(self.n_pairs), self.cap, _ptr(w), _stream()), "lgcn_widen_i32")
        return h[:P], w[:P]


def pairs_alloc(n_agt: int, n_scenes: int, cap: int, device):
    """Output / workspace buffers of pairs_build (allocate them on the stream that will consume the pairs when
    the search itself is enqueued on another stream)."""
    lib = L.load()
    i32 = dict(dtype=torch.int32, device=device)
    return (torch.empty(max(cap, 1), **i32), torch.empty(max(cap, 1), **i32), torch.empty(1, **i32),
            torch.empty(n_agt + 1, **i32), torch.empty(lib.lgcn_pairs_ws_elems(n_agt, n_scenes), **i32))


def pairs_build(agt_ctrs: torch.Tensor, agt_off: torch.Tensor, ctx_ctrs: torch.Tensor, ctx_off: torch.Tensor,
                dist_th: float, cap: int, legacy_offsets: bool = True, bufs=None) -> PairSet:
    lib = L.load()
    agt_ctrs = _dev(agt_ctrs, torch.float32, "agt_ctrs")
    ctx_ctrs = _dev(ctx_ctrs, torch.float32, "ctx_ctrs")
    agt_off = _dev(agt_off, torch.int32, "agt_off")
    ctx_off = _dev(ctx_off, torch.int32, "ctx_off")
    B = agt_off.numel() - 1
    if ctx_off.numel() != B + 1 or B < 1:
        raise L.LgcnError("pairs_build: offset tables must both have B+1 entries")
    T, S = agt_ctrs.shape[0], ctx_ctrs.shape[0]
    hi, wi, n_pairs, rowptr, ws = bufs if bufs is not None else pairs_alloc(T, B, cap, agt_ctrs.device)
    L.check(lib.lgcn_pairs_build(_ptr(agt_ctrs), _ptr(agt_off), _ptr(ctx_ctrs), _ptr(ctx_off), B, T, S,
                                 float(dist_th), int(bool(legacy_offsets)), _ptr(hi), _ptr(wi), cap,
                                 _ptr(n_pairs), _ptr(rowptr), _ptr(ws), _stream()), "lgcn_pairs_build")
    return PairSet(hi, wi, n_pairs, rowptr, cap, T, agt_ctrs, ctx_ctrs)


def pairs_build_multi(searches, legacy_offsets: bool = True, bufs=None) -> List["PairSet"]:
    """Up to four pair searches in the same three launches (the forward's A2M, M2A and A2A sets).  `searches`:
    tuples (agt_ctrs, agt_off, ctx_ctrs, ctx_off, dist_th, cap) as for pairs_build; `bufs`: pairs_alloc() per search."""
    lib = L.load()
    n = len(searches)
    if n < 1 or n > 4:
        raise L.LgcnError("pairs_build_multi: 1..4 searches")
    jobs, out, keep = _pairs_jobs(searches, legacy_offsets, bufs)
    L.check(lib.lgcn_pairs_build_multi(jobs, n, _stream()), "lgcn_pairs_build_multi")
    return out


def _pairs_jobs(searches, legacy_offsets, bufs):
    """ctypes job array + PairSets + tensors to keep alive for lgcn_pairs_build_multi / lgcn_index_build."""
    n = len(searches)
    jobs = (L.PairsJob * max(n, 1))()
    out, keep = [], []
    for k, (agt_ctrs, agt_off, ctx_ctrs, ctx_off, dist_th, cap) in enumerate(searches):
        agt_ctrs = _dev(agt_ctrs, torch.float32, "agt_ctrs")
        ctx_ctrs = _dev(ctx_ctrs, torch.float32, "ctx_ctrs")
        agt_off = _dev(agt_off, torch.int32, "agt_off")
        ctx_off = _dev(ctx_off, torch.int32, "ctx_off")
        B = agt_off.numel() - 1
        if ctx_off.numel() != B + 1 or B < 1:
            raise L.LgcnError("pair search: offset tables must both have B+1 entries")
        T, S = agt_ctrs.shape[0], ctx_ctrs.shape[0]
        hi, wi, n_pairs, rowptr, ws = bufs[k] if bufs is not None else pairs_alloc(T, B, cap, agt_ctrs.device)
        j = jobs[k]
        j.agt_ctrs, j.agt_off, j.ctx_ctrs, j.ctx_off = agt_ctrs.data_ptr(), agt_off.data_ptr(), ctx_ctrs.data_ptr(), ctx_off.data_ptr()
        j.n_scenes, j.legacy_offsets, j.n_agt, j.n_ctx, j.dist_th = B, int(bool(legacy_offsets)), T, S, float(dist_th)
        j.hi, j.wi, j.cap = hi.data_ptr(), wi.data_ptr(), cap
        j.n_pairs, j.rowptr, j.ws = n_pairs.data_ptr(), rowptr.data_ptr(), ws.data_ptr()
        keep += [agt_ctrs, ctx_ctrs, agt_off, ctx_off, ws]    # ws must outlive the launch (nothing else holds it)
        out.append(PairSet(hi, wi, n_pairs, rowptr, cap, T, agt_ctrs, ctx_ctrs))
    return jobs, out, keep


def index_counters(n_nodes: int, n_rel: int, device) -> torch.Tensor:
    """A zeroed counter buffer for index_build (lgcn_index_build needs it all zero and leaves it all zero): keep ONE
    per forward that can be in flight -- two launches sharing a buffer concurrently corrupt each other's counts."""
    lib = L.load()
    return torch.zeros(lib.lgcn_index_cnt_words(n_nodes, n_rel), dtype=torch.int64, device=device)


def index_fused_ok(n_nodes: int, n_rel: int, n_edges_total: int) -> bool:
    """Whether lgcn_index_build takes this size (else: graph_gather_indices + csr_build + pairs_build_multi)."""
    lib = L.load()
    return n_edges_total < (1 << 31) and lib.lgcn_csr_rowptr_elems(n_nodes, n_rel) <= (1 << 22)


def index_build(idx_local: torch.Tensor, seg_off: torch.Tensor, seg_base: torch.Tensor, rel_slices, n_nodes: int,
                searches=(), legacy_offsets: bool = True, bufs=None, cnt: Optional[torch.Tensor] = None,
                clear_word: Optional[torch.Tensor] = None):
    """graph_gather + CSR plan + the pair searches in four launches (lgcn_index_build).  rel_slices: per relation
    ((u_begin, u_end), (v_begin, v_end)) element ranges of idx_local.  cnt: index_counters() buffer owned by the caller
    (all zero; comes back all zero); None: a fresh one per call (one extra fill launch).  clear_word: an int32 device
    word that the first launch sets to 0 (the forward's range-guard flag).
    Returns (LanePlan, [PairSet])."""
    lib = L.load()
    idx_local = _dev(idx_local, torch.int64, "idx_local")
    seg_off = _dev(seg_off, torch.int64, "seg_off")
    seg_base = _dev(seg_base, torch.int64, "seg_base")
    n_rel = len(rel_slices)
    if n_rel < 1 or n_rel > L.MAX_REL or len(searches) > 4:
        raise L.LgcnError("index_build: 1..%d relations, <= 4 pair searches" % L.MAX_REL)
    dev = idx_local.device
    ne = [int(ub - ua) for (ua, ub), _ in rel_slices]
    nk1 = lib.lgcn_csr_rowptr_elems(n_nodes, n_rel)
    rowptr = torch.empty(nk1, dtype=torch.int32, device=dev)
    col = torch.empty(max(sum(ne), 1), dtype=torch.int32, device=dev)
    uv = torch.empty(max(2 * sum(ne), 2), dtype=torch.int32, device=dev)
    if cnt is None:
        cnt = index_counters(n_nodes, n_rel, dev)
    elif cnt.numel() < lib.lgcn_index_cnt_words(n_nodes, n_rel) or cnt.dtype != torch.int64 or cnt.device != dev:
        raise L.LgcnError("index_build: counter buffer of the wrong size / type / device")
    jobs, pairs, keep = _pairs_jobs(searches, legacy_offsets, bufs)
    p = L.Index()
    p.idx_local, p.n_elem = idx_local.data_ptr(), idx_local.numel()
    p.seg_off, p.seg_base, p.n_seg, p.n_rel = seg_off.data_ptr(), seg_base.data_ptr(), seg_base.numel(), n_rel
    for r, ((ua, ub), (va, vb)) in enumerate(rel_slices):
        if vb - va != ub - ua:
            raise L.LgcnError("index_build: u and v runs differ in length")
        p.u_off[r], p.v_off[r], p.n_edges[r] = ua, va, ub - ua
    p.n_nodes = n_nodes
    p.rowptr, p.col, p.cnt, p.uv = rowptr.data_ptr(), col.data_ptr(), cnt.data_ptr(), uv.data_ptr()
    p.jobs, p.n_jobs = C.cast(jobs, C.c_void_p).value if len(searches) else 0, len(searches)
    p.clear_word = 0 if clear_word is None else _dev(clear_word, torch.int32, "clear_word").data_ptr()
    L.check(lib.lgcn_index_build(C.byref(p), _stream()), "lgcn_index_build")
    plan = LanePlan(rowptr, col, n_rel, n_nodes, ne)
    plan.__dict__["_idx_keep"] = (cnt, uv, keep)       # alive until the plan is dropped (the launches are asynchronous)
    return plan, pairs


# ------------------------------------------------------------------ ActorNet's convolution block (row f1)
def conv_shape_ok(cin: int, cout: int, ks: int, stride: int, lin: int) -> bool:
    """Shapes lgcn_conv1d_gn takes (ActorNet's all do)."""
    if cin < 1 or cin > 128 or cout not in (32, 64, 128) or ks not in (1, 3) or stride not in (1, 2) or lin < 1:
        return False
    lout = (lin + 2 * ((ks - 1) // 2) - ks) // stride + 1
    return lout in (5, 10, 20)      # 16 / 8 / 4 actors per 80-row workgroup


def conv_packed(weight: torch.Tensor) -> torch.Tensor:
    """Packed image of a Conv1d weight [cout, cin, ks] for lgcn_conv1d_gn, cached on the parameter."""
    def make():
        lib = L.load()
        cout, cin, ks = weight.shape
        nbytes = lib.lgcn_conv_packed_bytes(cin, cout, ks)
        if nbytes < 0:
            raise L.LgcnError("conv_packed: unsupported Conv1d weight shape %s" % (tuple(weight.shape),))
        out = torch.empty(nbytes // 4, dtype=torch.int32, device=weight.device)
        w = _dev(weight.detach(), torch.float32, "weight")
        L.check(lib.lgcn_conv_pack_weight(_ptr(w), cin, cout, ks, _ptr(out), _stream()), "lgcn_conv_pack_weight")
        return out
    return _cached(weight, ("conv",), make)


def conv1d_gn(x: torch.Tensor, weight: torch.Tensor, stride: int, gamma, beta, eps: float, res: Optional[torch.Tensor] = None,
              res_up2: bool = False, relu: bool = False) -> torch.Tensor:
    """Conv1d (k = 1 / 3, padding (k - 1) / 2, no bias) + GroupNorm(1, C) + residual + ReLU on channels-last tensors in one
    launch (lgcn_conv1d_gn): x [A, L, Cin] -> [A, Lout, Cout]; res [A, Lout, Cout], or [A, Lout / 2, Cout] with res_up2."""
    lib = L.load()
    x = _dev(x, torch.float32, "x")
    A_, lin, cin = x.shape
    cout, cin_w, ks = weight.shape
    if cin_w != cin:
        raise L.LgcnError("conv1d_gn: weight does not match the input's channels")
    lout = (lin + 2 * ((ks - 1) // 2) - ks) // stride + 1
    out = torch.empty((A_, lout, cout), dtype=torch.float32, device=x.device)
    mode = 0
    if res is not None:
        res = _dev(res, torch.float32, "res")
        mode = 2 if res_up2 else 1
        if tuple(res.shape) != ((A_, lout // 2, cout) if res_up2 else (A_, lout, cout)):
            raise L.LgcnError("conv1d_gn: residual of the wrong shape")
    L.check(lib.lgcn_conv1d_gn(_ptr(x), A_, lin, cin, _ptr(conv_packed(weight)), cout, ks, stride,
                               _ptr(_dev(gamma.detach(), torch.float32, "gamma")), _ptr(_dev(beta.detach(), torch.float32, "beta")),
                               float(eps), _ptr(res), mode, int(bool(relu)), _ptr(out), _stream()), "lgcn_conv1d_gn")
    return out


def res1d_gn(x: torch.Tensor, block, second=None) -> torch.Tensor:
    """A whole layers.Res1d block (conv1 k3 + GN + ReLU + conv2 k3 + GN + shortcut [identity | conv k1 + GN] + ReLU) on a
    channels-last tensor x [A, L, Cin] in one launch (lgcn_res1d_gn) -> [A, Lout, C]; with `second` (a Res1d with the
    identity shortcut, C -> C) the two blocks run in the same launch (lgcn_res1d_pair_gn)."""
    lib = L.load()
    x = _dev(x, torch.float32, "x")
    A_, lin, cin = x.shape
    c1, c2, ds = block.conv1, block.conv2, block.downsample
    c, stride = c1.out_channels, c1.stride[0]
    if c1.in_channels != cin or c2.in_channels != c or c2.out_channels != c:
        raise L.LgcnError("res1d_gn: block does not match the input's channels")
    lout = (lin + 2 - 3) // stride + 1
    out = torch.empty((A_, lout, c), dtype=torch.float32, device=x.device)
    f32 = lambda t, n: _dev(t.detach(), torch.float32, n)
    wd = gd = bd = None
    if ds is not None:
        wd, gd, bd = conv_packed(ds[0].weight), f32(ds[1].weight, "gd"), f32(ds[1].bias, "bd")
    first = (_ptr(x), A_, lin, cin, c, stride, _ptr(conv_packed(c1.weight)), _ptr(f32(block.bn1.weight, "g1")),
             _ptr(f32(block.bn1.bias, "b1")), _ptr(conv_packed(c2.weight)), _ptr(f32(block.bn2.weight, "g2")),
             _ptr(f32(block.bn2.bias, "b2")), _ptr(wd), _ptr(gd), _ptr(bd))
    if second is None:
        L.check(lib.lgcn_res1d_gn(*first, float(block.bn1.eps), _ptr(out), _stream()), "lgcn_res1d_gn")
        return out
    q1, q2 = second.conv1, second.conv2
    if second.downsample is not None or q1.in_channels != c or q1.out_channels != c or q2.in_channels != c or q2.out_channels != c:
        raise L.LgcnError("res1d_gn: the chained block must be C -> C with the identity shortcut")
    L.check(lib.lgcn_res1d_pair_gn(*first, _ptr(conv_packed(q1.weight)), _ptr(f32(second.bn1.weight, "g1q")),
                                   _ptr(f32(second.bn1.bias, "b1q")), _ptr(conv_packed(q2.weight)),
                                   _ptr(f32(second.bn2.weight, "g2q")), _ptr(f32(second.bn2.bias, "b2q")),
                                   float(block.bn1.eps), _ptr(out), _stream()), "lgcn_res1d_pair_gn")
    return out


# ------------------------------------------------------------------ PredNet's tail (row f1)
def pred_reg(h: Sequence[torch.Tensor], w: Sequence[torch.Tensor], b: Sequence[torch.Tensor], ctrs: torch.Tensor,
             wd: torch.Tensor, bd: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """lgcn_pred_reg: reg [A, M, np2 / 2, 2] = w[m] h[m] + b[m] + ctr for the M heads, and AttDest's first layer
    hd [A M, 128] = relu(wd (ctr - reg[:, :, -1]) + bd), in one launch."""
    lib = L.load()
    M, A_ = len(h), h[0].shape[0]
    np2 = w[0].shape[0]
    q = L.PredReg()
    keep = []
    for m in range(M):
        hm, wm, bm = _dev(h[m], torch.float32, "h"), _dev(w[m].detach(), torch.float32, "w"), _dev(b[m].detach(), torch.float32, "b")
        if tuple(hm.shape) != (A_, C_FEAT) or tuple(wm.shape) != (np2, C_FEAT) or tuple(bm.shape) != (np2,):
            raise L.LgcnError("pred_reg: head %d has the wrong shape" % m)
        keep += [hm, wm, bm]
        q.h[m], q.w[m], q.b[m] = hm.data_ptr(), wm.data_ptr(), bm.data_ptr()
    ctrs, wd, bd = _dev(ctrs, torch.float32, "ctrs"), _dev(wd.detach(), torch.float32, "wd"), _dev(bd.detach(), torch.float32, "bd")
    if tuple(ctrs.shape) != (A_, 2) or tuple(wd.shape) != (C_FEAT, 2) or tuple(bd.shape) != (C_FEAT,):
        raise L.LgcnError("pred_reg: ctrs / dist weight of the wrong shape")
    reg = torch.empty((A_, M, np2 // 2, 2), dtype=torch.float32, device=ctrs.device)
    hd = torch.empty((A_ * M, C_FEAT), dtype=torch.float32, device=ctrs.device)
    keep += [ctrs, wd, bd]
    q.ctrs, q.wd, q.bd, q.reg, q.hd = (t.data_ptr() for t in (ctrs, wd, bd, reg, hd))
    q.n_act, q.n_mod, q.np2 = A_, M, np2
    L.check(lib.lgcn_pred_reg(C.byref(q), _stream()), "lgcn_pred_reg")
    return reg, hd


def pred_final(f: torch.Tensor, wc: torch.Tensor, bc: torch.Tensor, reg: torch.Tensor, rot: Optional[torch.Tensor] = None,
               orig: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """lgcn_pred_final: cls [A, M] (descending) = wc . f + bc and reg's modes in that order, taken to world coordinates
    when rot [A, 2, 2] / orig [A, 2] are given."""
    lib = L.load()
    A_, M, npred, _ = reg.shape
    f, reg = _dev(f, torch.float32, "f"), _dev(reg, torch.float32, "reg")
    wc, bc = _dev(wc.detach().reshape(-1), torch.float32, "wc"), _dev(bc.detach(), torch.float32, "bc")
    if tuple(f.shape) != (A_ * M, C_FEAT) or wc.numel() != C_FEAT or bc.numel() != 1:
        raise L.LgcnError("pred_final: score head of the wrong shape")
    if rot is not None:
        rot, orig = _dev(rot, torch.float32, "rot"), _dev(orig, torch.float32, "orig")
        if tuple(rot.shape) != (A_, 2, 2) or tuple(orig.shape) != (A_, 2):
            raise L.LgcnError("pred_final: rot / orig of the wrong shape")
    cls = torch.empty((A_, M), dtype=torch.float32, device=reg.device)
    out = torch.empty_like(reg)
    L.check(lib.lgcn_pred_final(_ptr(f), _ptr(wc), _ptr(bc), _ptr(reg), _ptr(rot), _ptr(orig), A_, M, npred,
                                _ptr(cls), _ptr(out), _stream()), "lgcn_pred_final")
    return cls, out


# ------------------------------------------------------------------ graph construction (row f3)
def dilated_nbrs(u: torch.Tensor, v: torch.Tensor, num_nodes: int, num_scales: int):
    """Scales 1 .. num_scales - 1 of a relation on the device (reference data.dilated_nbrs, data.py:520-534): the
    boolean powers A^(2^i) of the scale-0 adjacency by repeated squaring.  u, v: int64 device tensors (edge u <- v,
    i.e. row u, column v).  Returns a list of {"u", "v"} int64 device tensors, rows ascending, columns ascending
    within a row, every edge once.  Two host reads per scale (the sizes of the candidate and result arrays)."""
    lib = L.load()
    plan = csr_build([u], [v], num_nodes)              # one relation: key(n, 0) = n, a plain CSR by row
    n = plan.rowptr.numel() - 1                        # rows padded to a multiple of 16 (the padding rows are empty)
    dev = plan.rowptr.device
    rowptr, col = plan.rowptr, plan.col
    i32 = dict(dtype=torch.int32, device=dev)
    ws = torch.empty(max(lib.lgcn_scan_ws_elems(n + 1), 1), **i32)
    out = []
    for _ in range(1, num_scales):
        cand_ptr = torch.empty(n + 1, **i32)
        L.check(lib.lgcn_bool_square_bound(_ptr(rowptr), _ptr(col), n, _ptr(cand_ptr), _ptr(ws), _stream()),
                "lgcn_bool_square_bound")
        cand = torch.empty(max(int(cand_ptr[n].item()), 1), **i32)
        out_rowptr = torch.empty(n + 1, **i32)
        L.check(lib.lgcn_bool_square(_ptr(rowptr), _ptr(col), n, _ptr(cand_ptr), _ptr(cand), _ptr(out_rowptr), _ptr(ws),
                                     _stream()), "lgcn_bool_square")
        nnz = int(out_rowptr[n].item())
        out_col, out_row = torch.empty(max(nnz, 1), **i32), torch.empty(max(nnz, 1), **i32)
        L.check(lib.lgcn_bool_square_compact(_ptr(cand_ptr), _ptr(cand), _ptr(out_rowptr), n, _ptr(out_col), _ptr(out_row),
                                             _stream()), "lgcn_bool_square_compact")
        out.append({"u": out_row[:nnz].long(), "v": out_col[:nnz].long()})
        rowptr, col = out_rowptr, out_col
    return out


def cross_edges(ctrs: torch.Tensor, feats: torch.Tensor, lane_idcs: torch.Tensor, num_lanes: int,
                side_pairs: torch.Tensor, pre_pairs: torch.Tensor, suc_pairs: torch.Tensor, cross_dist: float):
    """Left (or right) node adjacency of one scene (lgcn_cross_edges; reference preprocess_data.py:287-392 without
    cross_angle): returns (u, v) int64 device tensors, u ascending."""
    lib = L.load()
    ctrs = _dev(ctrs, torch.float32, "ctrs")
    feats = _dev(feats, torch.float32, "feats")
    lane_idcs = _dev(lane_idcs, torch.int64, "lane_idcs")
    n = ctrs.shape[0]
    pairs = [_dev(t.reshape(-1, 2), torch.int64, "pairs") for t in (side_pairs, pre_pairs, suc_pairs)]
    dev = ctrs.device
    partner = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    mat = torch.empty(max(num_lanes * num_lanes, 1), dtype=torch.uint8, device=dev)
    L.check(lib.lgcn_cross_edges(_ptr(ctrs), _ptr(feats), _ptr(lane_idcs), n, num_lanes,
                                 _ptr(pairs[0]), pairs[0].shape[0], _ptr(pairs[1]), pairs[1].shape[0],
                                 _ptr(pairs[2]), pairs[2].shape[0], float(cross_dist), _ptr(mat), _ptr(partner),
                                 _stream()), "lgcn_cross_edges")
    partner = partner[:n]
    u = torch.nonzero(partner >= 0).reshape(-1)
    return u, partner[u].long()


# ------------------------------------------------------------------ weight packing
def _cached(weight: torch.Tensor, key, make):
    """Per-tensor-object cache, valid while (data_ptr, _version) are unchanged.  Living on the
    Parameter object itself, an entry can never be hit through a recycled device address of some
    other tensor.  (Writes through ``param.data`` do not bump ``_version``: call
    ``invalidate_packed(param)`` after such an edit.)"""
    cache = weight.__dict__.setdefault("_lgcn_cache", {})
    stamp = (weight.data_ptr(), weight._version, weight.device)
    hit = cache.get(key)
    if hit is not None and hit[0] == stamp:
        return hit[1]
    out = make()
    cache[key] = (stamp, out)
    return out


def invalidate_packed(weight: torch.Tensor):
    weight.__dict__.pop("_lgcn_cache", None)


# Every cached [128,128] image is also listed here, so that a training loop can rebuild all of them after an
# optimizer step with ONE launch (lgcn_pack_weight_batch) instead of ~400 single ones on first use.
_pack_jobs = []        # [weakref(weight), cache key, out, col0, transpose, mma]
_pack_tables = {}      # mma -> (job ids, device table [n,3] int64) of the last full refresh


def _register_pack(weight, key, out, col0, transpose, mma):
    import weakref
    _pack_jobs.append([weakref.ref(weight), key, out, col0, transpose, mma])


def refresh_packed() -> int:
    """Re-pack, in place, every registered image whose parameter changed since it was packed (optimizer step:
    reference train.py:190).  One launch per matrix-core mode in use.  Returns the number of images rebuilt."""
    if not _pack_jobs:
        return 0
    lib = L.load()
    stale = {}
    alive = []
    for job in _pack_jobs:
        w = job[0]()
        if w is None:
            continue
        entry = w.__dict__.get("_lgcn_cache", {}).get(job[1])
        if entry is None or entry[1] is not job[2] or not w.is_cuda:
            continue                     # cache entry dropped or replaced: nothing to refresh
        alive.append(job)
        if entry[0] != (w.data_ptr(), w._version, w.device):
            stale.setdefault(job[5], []).append((job, w))
    _pack_jobs[:] = alive
    n = 0
    for mma, items in stale.items():
        ids = tuple(id(j) for j, _ in items)
        hit = _pack_tables.get(mma)
        if hit is not None and hit[0] == ids and all(h == w.data_ptr() for h, (_, w) in zip(hit[2], items)):
            table = hit[1]
        else:
            rows = [[w.data_ptr() + 4 * j[3], j[2].data_ptr(), w.stride(0) | (int(j[4]) << 32)] for j, w in items]
            table = torch.tensor(rows, dtype=torch.int64).to(items[0][1].device)
            _pack_tables[mma] = (ids, table, [w.data_ptr() for _, w in items])
        L.check(lib.lgcn_pack_weight_batch(_ptr(table), len(items), mma, _stream()), "lgcn_pack_weight_batch")
        for j, w in items:
            w.__dict__["_lgcn_cache"][j[1]] = ((w.data_ptr(), w._version, w.device), j[2])
        n += len(items)
    return n


def packed(weight: torch.Tensor, col0: int = 0, k: Optional[int] = None) -> torch.Tensor:
    """MFMA-packed image of weight[:, col0:col0+k] ([128, k] slice of an nn.Linear weight); cached
    on the parameter and rebuilt only after the parameter changes."""
    lib = L.load()
    if weight.dim() != 2 or weight.shape[0] != C_FEAT:
        raise L.LgcnError("packed(): weight must be [128, K]")
    k = weight.shape[1] - col0 if k is None else k

    mma = _mma

    def make():
        w = _dev(weight.detach(), torch.float32, "weight")
        k_pad = (k + 7) // 8 * 8
        nbytes = lib.lgcn_packed_bytes(k_pad, mma)
        if nbytes < 0:
            raise L.LgcnError("packed(): K = %d is not supported in mma mode %d" % (k, mma))
        out = torch.empty(nbytes // 4, dtype=torch.float32, device=w.device)
        src = w[:, col0:]
        L.check(lib.lgcn_pack_weight(C.c_void_p(src.data_ptr()), w.stride(0), k, k_pad, mma, _ptr(out), _stream()),
                "lgcn_pack_weight")
        if k == C_FEAT:
            _register_pack(weight, ("pack", col0, k, mma), out, col0, 0, mma)
        return out

    return _cached(weight, ("pack", col0, k, mma), make)


def packed_t(weight: torch.Tensor, col0: int = 0) -> torch.Tensor:
    """MFMA image of the TRANSPOSE of the square block weight[:, col0:col0+128]: the backward of
    y = x W^T is dx = dy W, a Linear whose weight is W^T.  Cached like packed()."""
    lib = L.load()
    if weight.dim() != 2 or weight.shape[0] != C_FEAT or weight.shape[1] < col0 + C_FEAT:
        raise L.LgcnError("packed_t(): need a [128, >= col0+128] weight")
    mma = _mma

    def make():
        w = _dev(weight.detach(), torch.float32, "weight")
        out = torch.empty(lib.lgcn_packed_bytes(C_FEAT, mma) // 4, dtype=torch.float32, device=w.device)
        L.check(lib.lgcn_pack_weight_t(C.c_void_p(w[:, col0:].data_ptr()), w.stride(0), mma, _ptr(out), _stream()),
                "lgcn_pack_weight_t")
        _register_pack(weight, ("packT", col0, mma), out, col0, 1, mma)
        return out

    return _cached(weight, ("packT", col0, mma), make)


def cols4(weight: torch.Tensor, col0: int) -> torch.Tensor:
    """Contiguous [128, 4] copy of weight[:, col0:col0+4] (the 4 meta columns of A2M.meta), cached."""
    return _cached(weight, ("c4", col0), lambda: _dev(weight.detach(), torch.float32, "weight")[:, col0:col0 + 4].contiguous())


# ------------------------------------------------------------------ per-kernel timing hook
_timer = None


class kernel_timer:
    """``with ops.kernel_timer() as t:`` brackets every tagged launch with HIP events recorded on the
    stream the kernel is launched on (torch's current stream); ``t.summary()`` -> {tag: [ms, ...]}.
    Used by bench.py for the roofline of the dominant kernel; off (zero overhead) otherwise."""

    def __enter__(self):
        global _timer
        self.recs = []
        _timer = self
        return self

    def __exit__(self, *a):
        global _timer
        _timer = None

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for tag, e0, e1 in self.recs:
            out.setdefault(tag, []).append(e0.elapsed_time(e1))
        return out


class _Timed:
    def __init__(self, tag):
        self.on = _timer is not None and tag is not None
        self.tag = tag

    def __enter__(self):
        if self.on:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record()

    def __exit__(self, *a):
        if self.on:
            self.e1.record()
            _timer.recs.append((self.tag, self.e0, self.e1))


# ------------------------------------------------------------------ fp path
@dataclass
class RelSpec:
    src: torch.Tensor
    wp: Optional[torch.Tensor]      # packed weight (None for lgcn_wgrad)
    mode: int = L.REL_IDENT
    ridx: int = 0


def _agg_params(n_rows: int, rels: Sequence[RelSpec], flags: int, *, rowptr=None, col=None, n_rel_csr=0,
                gn1=None, wp2=None, gn2=None, res=None, x4=None, w4=None, out=None, out_pre=None, out_mid=None,
                out_pre2=None, eps=EPS, tile_rb=0, chain_u=None, chain_v=None):
    """Fill one lgcn_agg_mlp_t.  Returns (struct, tensors to keep alive until the launch is enqueued, out).
    chain_u = (wq packed, (gamma, beta), wu packed): also ReLU(GN_q(out wq^T)) wu^T -> out becomes (out, U[, V]);
    chain_v = wv packed: also out wv^T (lgcn.h: the chained outputs of a row block)."""
    if not rels or len(rels) > L.MAX_REL:
        raise L.LgcnError("agg_mlp: 1..%d relations" % L.MAX_REL)
    dev = rels[0].src.device
    p = L.AggMlp()
    p.n_rows, p.n_rel, p.n_rel_csr, p.flags, p.eps = n_rows, len(rels), n_rel_csr, flags, eps
    p.mma, p.tile_rb = _mma, tile_rb
    keep = []
    for i, r in enumerate(rels):
        s = _dev(r.src, torch.float32, "rel.src")
        keep += [s, r.wp]
        p.rel[i].src, p.rel[i].wp, p.rel[i].mode, p.rel[i].ridx = s.data_ptr(), r.wp.data_ptr(), r.mode, r.ridx
    p.rowptr = 0 if rowptr is None else rowptr.data_ptr()
    p.col = 0 if col is None else col.data_ptr()
    if w4 is not None:
        xa, xb, xc = (_dev(t, torch.float32, "x4") for t in x4)
        w4 = _dev(w4, torch.float32, "w4")
        keep += [xa, xb, xc, w4]
        p.x4_a, p.x4_b, p.x4_c, p.w4 = xa.data_ptr(), xb.data_ptr(), xc.data_ptr(), w4.data_ptr()
    if gn1 is not None:
        p.gn1_g, p.gn1_b = gn1[0].data_ptr(), gn1[1].data_ptr()
    if wp2 is not None:
        p.wp2 = wp2.data_ptr()
    if gn2 is not None:
        p.gn2_g, p.gn2_b = gn2[0].data_ptr(), gn2[1].data_ptr()
    if res is not None:
        res = _dev(res, torch.float32, "res")
        keep.append(res)
        p.res = res.data_ptr()
    if out is None:
        out = torch.empty((n_rows, C_FEAT), dtype=torch.float32, device=dev)
    p.out = out.data_ptr()
    p.out_pre = 0 if out_pre is None else out_pre.data_ptr()
    p.out_mid = 0 if out_mid is None else out_mid.data_ptr()
    p.out_pre2 = 0 if out_pre2 is None else out_pre2.data_ptr()
    if chain_u is not None or chain_v is not None:
        outs = [out]
        if chain_u is not None:
            wq, gq, wu = chain_u
            u = torch.empty((n_rows, C_FEAT), dtype=torch.float32, device=dev)
            keep += [wq, gq[0], gq[1], wu]
            p.ch_wq, p.ch_gq_g, p.ch_gq_b, p.ch_wu, p.ch_u_out = wq.data_ptr(), gq[0].data_ptr(), gq[1].data_ptr(), wu.data_ptr(), u.data_ptr()
            outs.append(u)
        if chain_v is not None:
            v = torch.empty((n_rows, C_FEAT), dtype=torch.float32, device=dev)
            keep.append(chain_v)
            p.ch_wv, p.ch_v_out = chain_v.data_ptr(), v.data_ptr()
            outs.append(v)
        out = tuple(outs)
    return p, keep, out


def agg_mlp(n_rows: int, rels: Sequence[RelSpec], flags: int, *, tag=None, **kw):
    """Fused aggregate -> GEMM -> GN -> ReLU -> GEMM -> GN -> +res -> ReLU row block (lgcn_agg_mlp).
    Keywords: rowptr, col, n_rel_csr, gn1, wp2, gn2, res, x4, w4, out, out_pre, out_mid, out_pre2, eps, tile_rb."""
    lib = L.load()
    p, keep, out = _agg_params(n_rows, rels, flags, **kw)
    with _Timed(tag):
        L.check(lib.lgcn_agg_mlp(C.byref(p), _stream()), "lgcn_agg_mlp")
    return out


def agg_mlp_pair(a: dict, b: dict, tag=None):
    """Two independent row blocks in one launch (lgcn_agg_mlp_pair).  a, b: keyword dicts of agg_mlp
    (n_rows, rels, flags, ...).  Returns (out_a, out_b)."""
    lib = L.load()
    pa, ka, oa = _agg_params(**a)
    pb, kb, ob = _agg_params(**b)
    with _Timed(tag):
        L.check(lib.lgcn_agg_mlp_pair(C.byref(pa), C.byref(pb), _stream()), "lgcn_agg_mlp_pair")
    return oa, ob


def agg_mlp_multi(problems: Sequence[dict], tag=None):
    """Up to 4 independent row blocks in one launch (lgcn_agg_mlp_multi).  problems: keyword dicts of agg_mlp.
    Returns their outputs in order (a tuple per problem with chained outputs)."""
    lib = L.load()
    built = [_agg_params(**kw) for kw in problems]
    arr = (C.POINTER(L.AggMlp) * len(built))(*[C.pointer(b[0]) for b in built])
    with _Timed(tag):
        L.check(lib.lgcn_agg_mlp_multi(arr, len(built), _stream()), "lgcn_agg_mlp_multi")
    return [b[2] for b in built]


# ------------------------------------------------------------------ LaneConv (gather-free, weight-stationary)
@dataclass
class LcPlan:
    """Work-item plan of lgcn_laneconv_fwd for one lane graph and one row-block geometry (include/lgcn.h)."""
    plan: torch.Tensor            # int32 words (lgcn_lc_plan_build)
    lane: LanePlan
    rows_per_block: int
    cap: int
    n_units: int
    gstart: List[int]             # unit groups: one workgroup per (row block, group)


def lc_config(mma: Optional[int] = None, variant: int = 0):
    """(rows_per_block, LDS source-row capacity) of a matrix mode (variant 0: shared, 1: tall, 2: short);
    None for LGCN_MMA_F32 and LGCN_MMA_BF16X3: those modes run the one-launch lgcn_agg_mlp LaneConv (the three-plane
    shapes of the weight-stationary kernel needed scratch and were slower: removed in round 3)."""
    lib = L.load()
    mma = _mma if mma is None else mma
    if mma in (L.MMA_F32, L.MMA_BF16X3):
        return None
    m, c = C.c_int32(), C.c_int32()
    L.check(lib.lgcn_lc_config(mma, variant, C.byref(m), C.byref(c)), "lgcn_lc_config")
    return m.value, c.value


def lc_groups(n_units: int, n_groups: int) -> List[int]:
    """Consecutive, near-equal unit groups: [0, ..., n_units]."""
    n_groups = max(1, min(n_groups, n_units))
    return [(g * n_units + n_groups - 1) // n_groups for g in range(n_groups)] + [n_units]


_cu_cache = {}


def cu_count(device) -> int:
    idx = torch.device(device).index
    idx = torch.cuda.current_device() if idx is None else idx
    if idx not in _cu_cache:
        _cu_cache[idx] = torch.cuda.get_device_properties(idx).multi_processor_count
    return _cu_cache[idx]


_lc_groups_forced = int(os.environ.get("LGCN_LC_GROUPS", "0"))
_lc_variant_forced = int(os.environ.get("LGCN_LC_VARIANT", "-1"))


def set_lc_variant(v: int):
    """Force the row-block shape of subsequently built LaneConv plans (0 shared, 1 tall, 2 short, -1 = pick by size)."""
    global _lc_variant_forced
    _lc_variant_forced = int(v)

_lc_impl = os.environ.get("LGCN_LANECONV", "tiled")
# Att layer for a given pair set: "split" (default) = U/V GEMMs + per-pair MLP (writes m [cap,128]) + segment-sum
# tail, three wide launches; "fused" = lgcn_att_fused, one launch per tile of targets that keeps the pair rows on
# the CU: no [cap,128] buffer (0.3-1.6 GB per call on large batches), measured 1.2-1.6x slower at S2.
_att_impl = os.environ.get("LGCN_ATT", "split")


def set_att_impl(name: str):
    global _att_impl
    if name not in ("fused", "split"):
        raise L.LgcnError("att impl must be 'fused' or 'split'")
    _att_impl = name


def att_impl() -> str:
    return _att_impl


def set_laneconv_impl(name: str):
    """"tiled" (lgcn_laneconv_fwd) or "fused" (one lgcn_agg_mlp launch per layer) for inference LaneConv layers."""
    global _lc_impl
    if name not in ("tiled", "fused"):
        raise L.LgcnError("laneconv impl must be 'tiled' or 'fused'")
    _lc_impl = name


def laneconv_impl() -> str:
    return _lc_impl


def set_lc_groups(n: int):
    """Force the number of unit groups of subsequently built LaneConv plans (0 = pick by CU count)."""
    global _lc_groups_forced
    _lc_groups_forced = int(n)


def lc_plan(lane: LanePlan, n_groups: Optional[int] = None, cap: Optional[int] = None,
            variant: Optional[int] = None) -> Optional[LcPlan]:
    """LaneConv work-item plan for the current matrix mode, built once per lane graph and cached on it.
    variant: 0 = shared (96-row blocks, two workgroups per CU), 1 = tall (192 rows, the whole CU), 2 = short (48 rows,
    one launch per layer, within the shared budget); default: short while its row blocks fit the CUs once, tall from
    two tall blocks per CU.
    n_groups: unit groups per row block (default 1: one launch per layer, no partial sums; more groups buy
    parallelism for a single small forward at the price of a second launch)."""
    if lc_config() is None:
        return None
    lib = L.load()
    if variant is None:
        variant = _lc_variant_forced
    if variant is None or variant < 0:
        cus = cu_count(lane.rowptr.device)
        m_tall, _ = lc_config(variant=1)
        m_short, _ = lc_config(variant=2)
        if (lane.n_nodes + m_short - 1) // m_short <= cus:
            # a small batch: one short row block per CU finishes the layer in one launch; the shape stays within
            # 128 VGPRs / 79 KB of LDS, so the layers of forwards in flight overlap two per CU
            variant = 2
        else:
            variant = 1 if (lane.n_nodes + m_tall - 1) // m_tall >= 2 * cus else 0
    M, cap_max = lc_config(variant=variant)
    cap = cap_max if cap is None else cap
    n_units = lane.n_rel + 1
    if n_groups is None:
        # one group (one launch per layer, no partial sums) once the row blocks alone fill the chip; small batches
        # buy parallelism with unit groups (S2, 108 short row blocks on 256 CUs: two groups)
        n_blocks = (lane.n_nodes + M - 1) // M
        n_groups = _lc_groups_forced if _lc_groups_forced > 0 else 1 if variant == 2 else max(1, min(4, round(cu_count(lane.rowptr.device) / max(n_blocks, 1))))
    gstart = lc_groups(n_units, n_groups)
    cache = lane.__dict__.setdefault("_lc", {})
    key = (M, cap, tuple(gstart))
    hit = cache.get(key)
    if hit is not None:
        return hit
    n_words = lib.lgcn_lc_plan_elems(lane.n_nodes, M, cap)
    if n_words < 0:
        raise L.LgcnError("lc_plan: bad geometry (n_nodes=%d, M=%d, cap=%d)" % (lane.n_nodes, M, cap))
    plan = torch.empty(max(n_words, 4), dtype=torch.int32, device=lane.rowptr.device)
    gs = (C.c_int32 * len(gstart))(*gstart)
    L.check(lib.lgcn_lc_plan_build(_ptr(lane.rowptr), _ptr(lane.col), lane.n_nodes, lane.n_rel, M, cap,
                                   len(gstart) - 1, gs, _ptr(plan), _stream()), "lgcn_lc_plan_build")
    out = LcPlan(plan, lane, M, cap, n_units, gstart)
    cache[key] = out
    return out


def lc_part(lcp: LcPlan) -> Optional[torch.Tensor]:
    """Partial-sum workspace of lgcn_laneconv_fwd (may be shared by consecutive layers on one stream); None for
    single-group plans, which need none."""
    lib = L.load()
    n = lib.lgcn_lc_part_elems(lcp.lane.n_nodes, lcp.rows_per_block, len(lcp.gstart) - 1)
    if n <= 0:
        return None
    return torch.empty(n, dtype=torch.float32, device=lcp.plan.device)


_lc_waves = int(os.environ.get("LGCN_LC_WAVES", "0"))      # 16: 16-wave workgroups for the short one-group shape


def set_lc_waves(n: int):
    """Waves per LaneConv workgroup of the short, one-group shape: 0 / 8 (default) or 16 (K split four ways: four waves
    per SIMD from one workgroup -- for ONE forward at a time; with several forwards in flight two 8-wave workgroups of
    different forwards share a CU instead)."""
    global _lc_waves
    if n not in (0, 8, 16):
        raise L.LgcnError("LaneConv workgroups have 8 or 16 waves")
    _lc_waves = n


def lc_waves() -> int:
    return _lc_waves


def laneconv_fwd(x: torch.Tensor, lcp: LcPlan, wps: Sequence[Optional[torch.Tensor]], gn1, wp2, gn2, eps=EPS,
                 part: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None, tag="laneconv",
                 waves: Optional[int] = None):
    """One LaneConv layer (lgcn_laneconv_fwd): wps[u] = packed weight of unit u (ctr, then the plan's relations;
    None for a relation without edges).  waves: None = the module setting (set_lc_waves) where the plan allows it."""
    lib = L.load()
    x = _dev(x, torch.float32, "x")
    if x.shape[0] != lcp.lane.n_nodes or len(wps) != lcp.n_units:
        raise L.LgcnError("laneconv_fwd: x / weights do not match the plan")
    p = L.LaneConv()
    p.n_rows, p.x = x.shape[0], x.data_ptr()
    for u, w in enumerate(wps):
        p.wp[u] = 0 if w is None else w.data_ptr()
    p.col, p.plan = lcp.lane.col.data_ptr(), lcp.plan.data_ptr()
    p.rows_per_block, p.cap, p.n_units, p.n_groups = lcp.rows_per_block, lcp.cap, lcp.n_units, len(lcp.gstart) - 1
    for g, v in enumerate(lcp.gstart):
        p.gstart[g] = v
    p.gn1_g, p.gn1_b, p.wp2, p.gn2_g, p.gn2_b = gn1[0].data_ptr(), gn1[1].data_ptr(), wp2.data_ptr(), gn2[0].data_ptr(), gn2[1].data_ptr()
    p.eps, p.mma = eps, _mma
    short = lc_config(variant=2)
    w = _lc_waves if waves is None else waves
    p.waves = 16 if (w == 16 and short is not None and lcp.rows_per_block == short[0] and len(lcp.gstart) == 2) else 0
    if part is None:
        part = lc_part(lcp)
    if out is None:
        out = torch.empty_like(x)
    p.part, p.out = (0 if part is None else part.data_ptr()), out.data_ptr()
    with _Timed(tag):
        L.check(lib.lgcn_laneconv_fwd(C.byref(p), _stream()), "lgcn_laneconv_fwd")
    return out


def mapnet_input(ctrs, feats, wa1, ba1, wpa2, gn_a, ws1, bs1, wps2, gn_s, eps=EPS):
    lib = L.load()
    ctrs = _dev(ctrs, torch.float32, "ctrs")
    feats = _dev(feats, torch.float32, "feats")
    n = ctrs.shape[0]
    out = torch.empty((n, C_FEAT), dtype=torch.float32, device=ctrs.device)
    L.check(lib.lgcn_mapnet_input(_ptr(ctrs), _ptr(feats), n, _ptr(wa1), _ptr(ba1), _ptr(wpa2), _ptr(gn_a[0]),
                                  _ptr(gn_a[1]), _ptr(ws1), _ptr(bs1), _ptr(wps2), _ptr(gn_s[0]), _ptr(gn_s[1]),
                                  eps, _mma, _ptr(out), _stream()), "lgcn_mapnet_input")
    return out


# Per-pair MLP of Att: "ws" (default in the 16-bit-plane modes) = lgcn_att_pairs_ws, both weights in registers;
# "stream" = lgcn_att_pairs, weight fragments streamed per 32-pair tile (the only one in f32).
# "wi" = lgcn_att_pairs_wi, wave-independent 16-pair blocks with both weights in LDS (f16x2 / bf16; the default there).
_att_pairs_impl = os.environ.get("LGCN_ATT_PAIRS", "wi")


def set_att_pairs_impl(name: str):
    global _att_pairs_impl
    if name not in ("wi", "ws", "stream"):
        raise L.LgcnError("att pairs impl must be 'wi', 'ws' or 'stream'")
    _att_pairs_impl = name


def att_pairs_impl() -> str:
    """The pair-MLP kernel of the current matrix mode: f32 has only "stream"; "wi" needs both weights in LDS (two or one
    16-bit plane: f16x2 / bf16), bf16x3 falls back to "ws"."""
    if _mma == L.MMA_F32:
        return "stream"
    if _att_pairs_impl == "wi" and _mma not in (L.MMA_F16X2, L.MMA_BF16):
        return "ws"
    return _att_pairs_impl


def packed_kperm(weight: torch.Tensor, col0: int = 0) -> torch.Tensor:
    """K-permuted MFMA image of weight[:, col0:col0+128] for lgcn_att_pairs_wi (lgcn_pack_weight_kperm); cached on the
    parameter like packed() (rebuilt on first use after the parameter changed)."""
    lib = L.load()
    mma = _mma

    def make():
        w = _dev(weight.detach(), torch.float32, "weight")
        out = torch.empty(lib.lgcn_packed_bytes(C_FEAT, mma) // 4, dtype=torch.float32, device=w.device)
        L.check(lib.lgcn_pack_weight_kperm(C.c_void_p(w[:, col0:].data_ptr()), w.stride(0), mma, _ptr(out), _stream()),
                "lgcn_pack_weight_kperm")
        return out

    return _cached(weight, ("packK", col0, mma), make)


def att_pairs(ps: PairSet, wd0, bd0, wpd2, gn_d, wpc0e, U, V, gn_c, m=None, eps=EPS, seg=0, tag="att_pairs"):
    """m [cap,128] of lgcn_att_pairs / lgcn_att_pairs_ws / lgcn_att_pairs_wi.  seg = 16 (ws / wi): per-target sums of
    16-aligned pieces at each piece's first row; hand m to agg_mlp as a REL_RANGE16 relation.
    wpd2 / wpc0e: (weight parameter, first column) -- packed here in the layout of the kernel in use -- or, for the ws /
    stream kernels, an image already made by packed()."""
    lib = L.load()
    impl = att_pairs_impl()
    if isinstance(wpd2, tuple) != isinstance(wpc0e, tuple):
        raise L.LgcnError("att_pairs: both weights as (parameter, column) or both packed")
    if isinstance(wpd2, tuple):
        pk = packed_kperm if impl == "wi" else (lambda w, c: packed(w, c, C_FEAT))
        wpd2, wpc0e = pk(*wpd2), pk(*wpc0e)
    elif impl == "wi":
        impl = "ws"          # images made by packed(): the kernels that read that layout
    if m is None:
        m = torch.empty((max(ps.cap, 1), C_FEAT), dtype=torch.float32, device=U.device)
    args = (_ptr(ps.agt_ctrs), _ptr(ps.ctx_ctrs), _ptr(ps.hi), _ptr(ps.wi), _ptr(ps.n_pairs),
            ps.cap, _ptr(wd0), _ptr(bd0), _ptr(wpd2), _ptr(gn_d[0]), _ptr(gn_d[1]), _ptr(wpc0e),
            _ptr(U), _ptr(V), _ptr(gn_c[0]), _ptr(gn_c[1]), eps, _mma)
    with _Timed(tag):
        if impl == "wi":
            rc = lib.lgcn_att_pairs_wi(*args, seg, _ptr(m), _stream())
        elif impl == "ws":
            rc = lib.lgcn_att_pairs_ws(*args, seg, _ptr(m), _stream())
        elif seg != 0:
            raise L.LgcnError("att_pairs: seg needs the weight-stationary kernel (16-bit-plane modes)")
        else:
            rc = lib.lgcn_att_pairs(*args, _ptr(m), _stream())
    L.check(rc, "lgcn_att_pairs")
    return m


def att_targets_per_block(n_agt: int, device) -> int:
    """Target rows per workgroup of att_fused: as many as still give the chip ~a workgroup per CU (8..32)."""
    forced = int(os.environ.get("LGCN_ATT_TT", "0"))        # tests / tuning
    if forced:
        return forced
    per = n_agt / max(1, cu_count(device))
    return 32 if per >= 24 else 16 if per >= 12 else 8      # lgcn_att_fused also takes 4


def att_fused(agts, ps: PairSet, V, wq, gn_q, wc0q, wd0, bd0, wd2, gn_d, wc0e, gn_c, wagt, wc1, gn_n, wlin, gn_l,
              eps=EPS, targets_per_block: Optional[int] = None, tag="att_fused"):
    """One Att layer for a given pair set (lgcn_att_fused): packed weights w*, (gamma, beta) pairs gn_*."""
    lib = L.load()
    agts = _dev(agts, torch.float32, "agts")
    out = torch.empty_like(agts)
    p = L.AttFused()
    p.agts, p.n_agt, p.agt_ctrs, p.ctx_ctrs = agts.data_ptr(), agts.shape[0], ps.agt_ctrs.data_ptr(), ps.ctx_ctrs.data_ptr()
    p.hi, p.wi, p.rowptr, p.cap = ps.hi.data_ptr(), ps.wi.data_ptr(), ps.rowptr.data_ptr(), ps.cap
    p.wpq, p.gq, p.bq, p.wpc0q = wq.data_ptr(), gn_q[0].data_ptr(), gn_q[1].data_ptr(), wc0q.data_ptr()
    p.wd0, p.bd0, p.wpd2, p.gd, p.btd = wd0.data_ptr(), bd0.data_ptr(), wd2.data_ptr(), gn_d[0].data_ptr(), gn_d[1].data_ptr()
    p.wpc0e, p.V, p.gc, p.btc = wc0e.data_ptr(), V.data_ptr(), gn_c[0].data_ptr(), gn_c[1].data_ptr()
    p.wpagt, p.wpc1, p.gn, p.bn = wagt.data_ptr(), wc1.data_ptr(), gn_n[0].data_ptr(), gn_n[1].data_ptr()
    p.wplin, p.gl, p.bl = wlin.data_ptr(), gn_l[0].data_ptr(), gn_l[1].data_ptr()
    p.eps, p.mma = eps, _mma
    p.targets_per_block = targets_per_block or att_targets_per_block(agts.shape[0], agts.device)
    p.out = out.data_ptr()
    with _Timed(tag):
        L.check(lib.lgcn_att_fused(C.byref(p), _stream()), "lgcn_att_fused")
    return out


def pred_loss_fwd(cls, reg, gt, has, cfg):
    """lgcn_pred_loss_fwd: (sums [2] fp32 = cls_loss, reg_loss; counts [2] int32 = num_cls, num_reg; sel [A] int32)."""
    lib = L.load()
    A, M = cls.shape
    T = reg.shape[2]
    dev = cls.device
    sums = torch.empty(2, dtype=torch.float32, device=dev)
    counts = torch.empty(2, dtype=torch.int32, device=dev)
    sel = torch.empty(max(A, 1), dtype=torch.int32, device=dev)
    L.check(lib.lgcn_pred_loss_fwd(_ptr(cls), _ptr(reg), _ptr(gt), _ptr(has), A, M, T, cfg["cls_th"], cfg["cls_ignore"], cfg["mgn"],
                                   cfg["cls_coef"], cfg["reg_coef"], _ptr(sums), _ptr(counts), _ptr(sel), _stream()), "lgcn_pred_loss_fwd")
    return sums, counts, sel


def pred_loss_bwd(cls, reg, gt, has, cfg, sel, g_cls, g_reg):
    lib = L.load()
    A, M = cls.shape
    T = reg.shape[2]
    dcls, dreg = torch.empty_like(cls), torch.empty_like(reg)
    L.check(lib.lgcn_pred_loss_bwd(_ptr(cls), _ptr(reg), _ptr(gt), _ptr(has), A, M, T, cfg["cls_coef"], cfg["reg_coef"], _ptr(sel),
                                   _ptr(g_cls), _ptr(g_reg), _ptr(dcls), _ptr(dreg), _stream()), "lgcn_pred_loss_bwd")
    return dcls, dreg


# ------------------------------------------------------------------ backward building blocks
def gn_fwd(x, gn=None, res=None, relu=False, eps=EPS):
    """out = [ReLU](GroupNorm(1,128)(x) [+ res]) as a stand-alone row kernel."""
    lib = L.load()
    x = _dev(x, torch.float32, "x")
    out = torch.empty_like(x)
    g, b = (gn[0], gn[1]) if gn is not None else (None, None)
    res = None if res is None else _dev(res, torch.float32, "res")
    L.check(lib.lgcn_gn_fwd(_ptr(x), _ptr(g), _ptr(b), _ptr(res), x.shape[0], eps, int(bool(relu)), _ptr(out), _stream()),
            "lgcn_gn_fwd")
    return out


def gn_cl(x, gamma, beta, eps=EPS, res=None, relu=False, res_up2=False):
    """out = [ReLU](GroupNorm(1 group over (C, L))(x) [+ res]) in one launch: ActorNet's conv norms.
    x: [n, C, L] contiguous, or [n, C, 1, L] in torch.channels_last (memory [n, L, C]: the layout MIOpen's
    convolutions take without transposes); out has x's shape and layout.  res_up2: res is at half length and is
    upsampled x2 (linear, align_corners=False) on the fly (FPN top-down step)."""
    lib = L.load()
    if not x.is_cuda or x.dtype != torch.float32:
        raise L.LgcnError("gn_cl: x must be a float32 CUDA tensor")
    cl = x.dim() == 4
    if cl:
        if x.shape[2] != 1 or not x.is_contiguous(memory_format=torch.channels_last):
            raise L.LgcnError("gn_cl: 4-D input must be [n, C, 1, L] in channels_last")
        n, C_, L_ = x.shape[0], x.shape[1], x.shape[3]
    elif x.dim() == 3:
        x = x.contiguous()
        n, C_, L_ = x.shape
    else:
        raise L.LgcnError("gn_cl: x must be [n, C, L] or [n, C, 1, L]")
    if res is not None:
        want = list(x.shape)
        if res_up2:
            want[-1] //= 2
        if list(res.shape) != want or (res_up2 and L_ % 2) or not res.is_cuda or res.dtype != torch.float32:
            raise L.LgcnError("gn_cl: res has shape %s, expected %s" % (tuple(res.shape), tuple(want)))
        if cl and not res.is_contiguous(memory_format=torch.channels_last):
            raise L.LgcnError("gn_cl: res must be channels_last like x")
        if not cl:
            res = res.contiguous()
    out = torch.empty_like(x)
    gamma, beta = _dev(gamma.detach(), torch.float32, "gamma"), _dev(beta.detach(), torch.float32, "beta")
    L.check(lib.lgcn_gn_cl(_ptr(x), n, C_, L_, _ptr(gamma), _ptr(beta), float(eps), _ptr(res),
                           int(bool(res_up2 and res is not None)), int(bool(relu)), int(cl), _ptr(out), _stream()),
            "lgcn_gn_cl")
    return out


def gn_cl_bwd(dy, x, post, gamma, eps=EPS, want_g=False):
    """Backward of gn_cl on [n, C, L]: (dx, g | None, dgamma, dbeta); post = forward output when it applied a ReLU."""
    lib = L.load()
    dy, x = _dev(dy, torch.float32, "dy").contiguous(), _dev(x, torch.float32, "x").contiguous()
    n, C_, L_ = x.shape
    post = None if post is None else _dev(post, torch.float32, "post").contiguous()
    gamma = _dev(gamma.detach(), torch.float32, "gamma")
    dx = torch.empty_like(x)
    g = torch.empty_like(x) if want_g else None
    part = torch.empty((n, 2, C_), dtype=torch.float32, device=x.device)
    L.check(lib.lgcn_gn_cl_bwd(_ptr(dy), _ptr(x), _ptr(post), _ptr(gamma), n, C_, L_, float(eps), _ptr(dx), _ptr(g),
                               _ptr(part), _stream()), "lgcn_gn_cl_bwd")
    sums = part.sum(0)
    return dx, g, sums[0], sums[1]


def gn_bwd(dy, x, post, gamma, eps=EPS, want_g=False):
    """Backward of out = [ReLU](GN(x) [+res]): returns (dx, g, dgamma, dbeta); g = dy masked by post > 0
    (the gradient into `res`), None unless want_g.  gamma None: mask only."""
    lib = L.load()
    dy = _dev(dy, torch.float32, "dy")
    n = dy.shape[0]
    dx = torch.empty_like(dy)
    g = torch.empty_like(dy) if want_g else None
    if gamma is None:
        L.check(lib.lgcn_gn_bwd(_ptr(dy), None, _ptr(post), None, n, eps, _ptr(dx), _ptr(g), None, None, None, _stream()),
                "lgcn_gn_bwd")
        return dx, g, None, None
    x = _dev(x, torch.float32, "x")
    dgamma = torch.empty(C_FEAT, dtype=torch.float32, device=dy.device)
    dbeta = torch.empty_like(dgamma)
    part = torch.empty(2 * ((n + 31) // 32) * C_FEAT, dtype=torch.float32, device=dy.device)
    L.check(lib.lgcn_gn_bwd(_ptr(dy), _ptr(x), _ptr(post), _ptr(gamma), n, eps, _ptr(dx), _ptr(g), _ptr(dgamma),
                            _ptr(dbeta), _ptr(part), _stream()), "lgcn_gn_bwd")
    return dx, g, dgamma, dbeta


def wgrad(n_rows: int, rels: Sequence[RelSpec], dT: torch.Tensor, *, rowptr=None, col=None, n_rel_csr=0,
          n_chunks: Optional[int] = None) -> torch.Tensor:
    """dW[r] = dT^T (G_r src_r) for every relation: [n_rel,128,128] fp32.  The rows are cut into n_chunks
    partial sums per relation (default: ~512 workgroups in all, two per CU; a single-relation launch over the
    67 k pairs of A2A took 557 us on 16 workgroups)."""
    lib = L.load()
    dT = _dev(dT, torch.float32, "dT")
    p = L.AggMlp()
    p.n_rows, p.n_rel, p.n_rel_csr = n_rows, len(rels), n_rel_csr
    keep = []
    for i, r in enumerate(rels):
        s = _dev(r.src, torch.float32, "rel.src")
        keep.append(s)
        p.rel[i].src, p.rel[i].mode, p.rel[i].ridx = s.data_ptr(), r.mode, r.ridx
    p.rowptr = 0 if rowptr is None else rowptr.data_ptr()
    p.col = 0 if col is None else col.data_ptr()
    if n_chunks is None:
        n_chunks = max(1, 512 // len(rels))
    n_chunks = max(1, min(n_chunks, (n_rows + 31) // 32, 1024))
    dW = torch.empty((len(rels), C_FEAT, C_FEAT), dtype=torch.float32, device=dT.device)
    part = torch.empty(len(rels) * n_chunks * C_FEAT * C_FEAT, dtype=torch.float32, device=dT.device)
    L.check(lib.lgcn_wgrad(C.byref(p), _ptr(dT), _ptr(dW), _ptr(part), n_chunks, _stream()), "lgcn_wgrad")
    return dW


def gather_rows(src, idx, n_dev, cap):
    lib = L.load()
    src = _dev(src, torch.float32, "src")
    out = torch.empty((max(cap, 1), C_FEAT), dtype=torch.float32, device=src.device)
    L.check(lib.lgcn_gather_rows(_ptr(src), _ptr(idx), _ptr(n_dev), cap, _ptr(out), _stream()), "lgcn_gather_rows")
    return out[:cap]


def gather_sum(src, rowptr, col, n_rows):
    lib = L.load()
    src = _dev(src, torch.float32, "src")
    out = torch.empty((n_rows, C_FEAT), dtype=torch.float32, device=src.device)
    L.check(lib.lgcn_gather_sum(_ptr(src), _ptr(rowptr), _ptr(col), n_rows, _ptr(out), _stream()), "lgcn_gather_sum")
    return out


def pair_add(c, U, hi, V, wi, n_dev, cap):
    lib = L.load()
    c = _dev(c, torch.float32, "c")
    out = torch.empty_like(c)
    L.check(lib.lgcn_pair_add(_ptr(c), _ptr(U), _ptr(hi), _ptr(V), _ptr(wi), _ptr(n_dev), cap, _ptr(out), _stream()),
            "lgcn_pair_add")
    return out
