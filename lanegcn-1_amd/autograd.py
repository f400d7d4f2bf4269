"""Differentiable versions of the hot-path row blocks (training path).

Forward = the fused HIP kernels of ops.py (saving the pre-normalisation tensors); backward = the same
kernels on transposed plans / transposed weights plus three backward kernels:
  row-GEMMs of the backward  -> lgcn_agg_mlp  (dx = dy W is a Linear with weight W^T; the transpose of a
                                 gather-by-destination is a gather-by-source on the transposed plan)
  GroupNorm / ReLU backward  -> lgcn_gn_bwd   (deterministic dgamma / dbeta)
  weight gradients           -> lgcn_wgrad    (dW_r = dT^T (G_r src_r), fp32-input MFMA)
Only the K = 2 / K = 4 input Linears (nn.Linear(2,128) of the stems, the 4 meta columns) stay on stock
ATen ops in the training path: they are [N,2]-shaped, not 128-d contractions.
"""
from dataclasses import dataclass
from typing import List, Optional

import torch
from torch.autograd import Function

from . import _lib as L
from . import ops

C_FEAT = ops.C_FEAT


@dataclass
class Rel:
    """One relation of a row block: which source tensor / weight (indices into the Function's tensor
    arguments), how it is gathered and (for weight slices) the first input column of the 128-wide block."""
    src: int                    # index into `srcs`
    w: int                      # index into `weights`
    mode: int = L.REL_IDENT
    ridx: int = 0
    col0: int = 0


@dataclass
class BlockSpec:
    n_rows: int
    rels: List[Rel]
    gn: bool = False
    relu: bool = False
    has_res: bool = False
    eps: float = ops.EPS
    plan: Optional[ops.LanePlan] = None        # CSR relations: plan by destination ...
    plan_t: Optional[ops.LanePlan] = None      # ... and by source (for d src)
    rowptr: Optional[torch.Tensor] = None      # RANGE relation: segments [n_rows+1]
    seg_ids: Optional[torch.Tensor] = None     # RANGE relation: segment id of every source row (int32)
    n_seg_rows: Optional[torch.Tensor] = None  # device count of valid source rows (int32 [1])
    tag: Optional[str] = None


def _fwd_rels(spec: BlockSpec, srcs, weights):
    return [ops.RelSpec(srcs[r.src], ops.packed(weights[r.w], r.col0, C_FEAT), r.mode, r.ridx) for r in spec.rels]


def _csr_kw(spec: BlockSpec, plan):
    if plan is not None:
        return dict(rowptr=plan.rowptr, col=plan.col, n_rel_csr=plan.n_rel)
    if spec.rowptr is not None:
        return dict(rowptr=spec.rowptr)
    return {}


def _stage_backward(spec: BlockSpec, srcs, weights, dT, need_src, need_w, res_grad=None):
    """Gradients of T = sum_r (G_r src_r) W_r^T given dT.  Returns (d_srcs list, d_weights list)."""
    d_srcs: List[Optional[torch.Tensor]] = [None] * len(srcs)
    d_ws: List[Optional[torch.Tensor]] = [None] * len(weights)
    # ---- d src: one launch per source for the IDENT / CSR relations, gather for RANGE
    for si in range(len(srcs)):
        if not need_src[si]:
            continue
        same = [r for r in spec.rels if r.src == si]
        lin = [r for r in same if r.mode != L.REL_RANGE]
        acc = None
        if lin:
            rels = []
            for r in lin:
                wp = ops.packed_t(weights[r.w], r.col0)
                rels.append(ops.RelSpec(dT, wp, L.REL_CSR if r.mode == L.REL_CSR else L.REL_IDENT, r.ridx))
            flags = L.F_RES if (res_grad is not None and si == 0) else 0
            acc = ops.agg_mlp(srcs[si].shape[0], rels, flags, res=res_grad if flags else None,
                              **_csr_kw(spec, spec.plan_t if any(r.mode == L.REL_CSR for r in lin) else None))
        for r in same:
            if r.mode == L.REL_RANGE:   # d src[p] = (dT W_r)[seg(p)]
                tmp = ops.agg_mlp(dT.shape[0], [ops.RelSpec(dT, ops.packed_t(weights[r.w], r.col0))], 0)
                rows = srcs[si].shape[0]
                g = ops.gather_rows(tmp, spec.seg_ids, spec.n_seg_rows, rows)
                acc = g if acc is None else acc + g
        d_srcs[si] = acc
    # ---- d W
    if any(need_w):
        rels = [ops.RelSpec(srcs[r.src], None, r.mode, r.ridx) for r in spec.rels]
        dW = ops.wgrad(spec.n_rows, rels, dT, **_csr_kw(spec, spec.plan))
        for i, r in enumerate(spec.rels):
            if not need_w[r.w]:
                continue
            w = weights[r.w]
            if w.shape[1] == C_FEAT:
                d_ws[r.w] = dW[i] if d_ws[r.w] is None else d_ws[r.w] + dW[i]
            else:                           # a 128-column block of a wider weight (ctx.0 [128,384], meta [128,132])
                if d_ws[r.w] is None:
                    d_ws[r.w] = torch.zeros_like(w)
                d_ws[r.w][:, r.col0:r.col0 + C_FEAT] += dW[i]
    return d_srcs, d_ws


class RowBlockFn(Function):
    """out = [ReLU]( [GN]( sum_r (G_r src_r) W_r^T ) [+ res] )  -- layers.Linear, Att.query / agt+ctx.1 tail,
    the pure Linear stages (gn = relu = False)."""

    @staticmethod
    def forward(ctx, spec: BlockSpec, n_src: int, n_w: int, *tensors):
        srcs, weights = list(tensors[:n_src]), list(tensors[n_src:n_src + n_w])
        gn_w, gn_b, res = tensors[n_src + n_w:n_src + n_w + 3]
        flags = (L.F_GN1 if spec.gn else 0) | (L.F_RELU1 if spec.relu else 0) | (L.F_RES if spec.has_res else 0)
        need_pre = spec.gn or spec.relu
        pre = torch.empty((spec.n_rows, C_FEAT), dtype=torch.float32, device=srcs[0].device) if need_pre else None
        out = ops.agg_mlp(spec.n_rows, _fwd_rels(spec, srcs, weights), flags,
                          gn1=(gn_w, gn_b) if spec.gn else None, res=res if spec.has_res else None, out_pre=pre,
                          eps=spec.eps, tag=spec.tag, **_csr_kw(spec, spec.plan))
        ctx.spec, ctx.n_src, ctx.n_w = spec, n_src, n_w
        ctx.save_for_backward(*srcs, *weights, *(t for t in (gn_w, pre, out) if t is not None))
        ctx.has = (gn_w is not None, pre is not None)
        return out

    @staticmethod
    def backward(ctx, d_out):
        spec, n_src, n_w = ctx.spec, ctx.n_src, ctx.n_w
        saved = list(ctx.saved_tensors)
        srcs, weights = saved[:n_src], saved[n_src:n_src + n_w]
        rest = saved[n_src + n_w:]
        gn_w = rest.pop(0) if ctx.has[0] else None
        pre = rest.pop(0) if ctx.has[1] else None
        out = rest.pop(0)
        d_out = d_out.contiguous()
        ni = ctx.needs_input_grad          # (spec, n_src, n_w, *tensors)
        need_src = [ni[3 + i] for i in range(n_src)]
        need_w = [ni[3 + n_src + i] for i in range(n_w)]
        d_gw = d_gb = d_res = None
        if spec.gn or spec.relu:
            dT, g, d_gw, d_gb = ops.gn_bwd(d_out, pre, out if spec.relu else None, gn_w if spec.gn else None,
                                           eps=spec.eps, want_g=spec.has_res)
            d_res = g if spec.has_res else None
        else:
            dT = d_out
            d_res = d_out if spec.has_res else None
        with ops.backward_mma():
            d_srcs, d_ws = _stage_backward(spec, srcs, weights, dT, need_src, need_w)
        return (None, None, None, *d_srcs, *d_ws, d_gw, d_gb, d_res)


class LaneConvFn(Function):
    """One fused LaneConv layer (lanegcn.py:331-362): X' = ReLU(GN2(ReLU(GN1(sum_r (G_r X) W_r^T)) W2^T) + X).
    Forward is the single fused launch of inference (saving T, Y, Z); backward is composed."""

    @staticmethod
    def forward(ctx, spec: BlockSpec, feat, gn1_w, gn1_b, w2, gn2_w, gn2_b, *weights):
        N = spec.n_rows
        T, Y, Z = (torch.empty((N, C_FEAT), dtype=torch.float32, device=feat.device) for _ in range(3))
        flags = L.F_GN1 | L.F_RELU1 | L.F_GEMM2 | L.F_GN2 | L.F_RES | L.F_RELU2
        out = ops.agg_mlp(N, _fwd_rels(spec, [feat], list(weights)), flags, gn1=(gn1_w, gn1_b), wp2=ops.packed(w2),
                          gn2=(gn2_w, gn2_b), res=feat, out_pre=T, out_mid=Y, out_pre2=Z, eps=spec.eps, tag="laneconv",
                          **_csr_kw(spec, spec.plan))
        ctx.spec = spec
        ctx.save_for_backward(feat, gn1_w, w2, gn2_w, T, Y, Z, out, *weights)
        return out

    @staticmethod
    def backward(ctx, d_out):
        spec = ctx.spec
        feat, gn1_w, w2, gn2_w, T, Y, Z, out, *weights = ctx.saved_tensors
        ni = ctx.needs_input_grad
        N = spec.n_rows
        # out = ReLU(GN2(Z) + X)
        dZ, g2, d_g2w, d_g2b = ops.gn_bwd(d_out.contiguous(), Z, out, gn2_w, eps=spec.eps, want_g=True)
        # Z = Y W2^T
        with ops.backward_mma():
            dY = ops.agg_mlp(N, [ops.RelSpec(dZ, ops.packed_t(w2))], 0)
        d_w2 = ops.wgrad(N, [ops.RelSpec(Y, None)], dZ)[0] if ni[4] else None
        # Y = ReLU(GN1(T))
        dT, _, d_g1w, d_g1b = ops.gn_bwd(dY, T, Y, gn1_w, eps=spec.eps)
        # T = sum_r (G_r X) W_r^T ; the residual branch adds g2 to dX inside the same launch
        need_w = [ni[7 + i] for i in range(len(weights))]
        with ops.backward_mma():
            d_srcs, d_ws = _stage_backward(spec, [feat], list(weights), dT, [ni[1]], need_w, res_grad=g2)
        return (None, d_srcs[0], d_g1w, d_g1b, d_w2, d_g2w, d_g2b, *d_ws)


class GNActFn(Function):
    """out = [ReLU](GN(x) [+ res]) on rows (stand-alone; the per-pair composition of Att)."""

    @staticmethod
    def forward(ctx, x, gn_w, gn_b, res, relu: bool, eps: float):
        out = ops.gn_fwd(x, (gn_w, gn_b) if gn_w is not None else None, res, relu, eps)
        ctx.relu, ctx.eps, ctx.has_res, ctx.has_gn = relu, eps, res is not None, gn_w is not None
        ctx.save_for_backward(x, out, *([gn_w] if gn_w is not None else []))
        return out

    @staticmethod
    def backward(ctx, d_out):
        x, out, *gw = ctx.saved_tensors
        dx, g, d_gw, d_gb = ops.gn_bwd(d_out.contiguous(), x, out if ctx.relu else None, gw[0] if ctx.has_gn else None,
                                       eps=ctx.eps, want_g=ctx.has_res)
        return dx, d_gw, d_gb, (g if ctx.has_res else None), None, None


class GNCLFn(Function):
    """out = [ReLU](GroupNorm(1 group over (C, L))(x) [+ res]) on [n, C, L]: ActorNet's conv norms (one launch
    forward, one backward, instead of ~30 ATen launches of the explicit mean / var formula)."""

    @staticmethod
    def forward(ctx, x, gn_w, gn_b, res, relu: bool, eps: float):
        x = x.contiguous()
        out = ops.gn_cl(x, gn_w, gn_b, eps, res=res, relu=relu)
        ctx.relu, ctx.eps, ctx.has_res = relu, eps, res is not None
        ctx.save_for_backward(x, out, gn_w)
        return out

    @staticmethod
    def backward(ctx, d_out):
        x, out, gw = ctx.saved_tensors
        dx, g, d_gw, d_gb = ops.gn_cl_bwd(d_out, x, out if ctx.relu else None, gw, eps=ctx.eps, want_g=ctx.has_res)
        return dx, d_gw, d_gb, (g if ctx.has_res else None), None, None


def gn_cl_act(x, gn, relu=False, res=None):
    return GNCLFn.apply(x, gn.weight, gn.bias, res, relu, gn.eps)


class PairAddFn(Function):
    """out[p] = c[p] + U[hi[p]] + V[wi[p]] (the hoisted query / context terms of lanegcn.py:696-699)."""

    @staticmethod
    def forward(ctx, c, U, V, pairs):
        P = c.shape[0]
        out = ops.pair_add(c, U, pairs.hi, V, pairs.wi, pairs.n_pairs, P)
        ctx.pairs, ctx.n_u, ctx.n_v = pairs, U.shape[0], V.shape[0]
        return out

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        ps = ctx.pairs
        dU = ops.gather_sum(g, ps.rowptr, None, ctx.n_u) if ctx.needs_input_grad[1] else None       # sorted by hi
        dV = None
        if ctx.needs_input_grad[2]:
            rp, col = ps.csr_by_wi(ctx.n_v)
            dV = ops.gather_sum(g, rp, col, ctx.n_v)
        return g, dU, dV, None


class GatherSumFn(Function):
    """out[n] = sum of src rows over a CSR (rowptr, col) by destination: the differentiable index_add_ of a gathered
    tensor (reference lanercnn.py:343: out.index_add_(0, v, agt_fc(agt[u]))).  Backward = the same gather over the
    transposed CSR (plan_t: rows = sources)."""

    @staticmethod
    def forward(ctx, src, plan, plan_t, n_rows: int):
        ctx.plan_t, ctx.n_src = plan_t, src.shape[0]
        return ops.gather_sum(src.contiguous(), plan.rowptr, plan.col, n_rows)

    @staticmethod
    def backward(ctx, g):
        return ops.gather_sum(g.contiguous(), ctx.plan_t.rowptr, ctx.plan_t.col, ctx.n_src), None, None, None


# ------------------------------------------------------------------ convenience wrappers
def row_block(srcs, weights, rels, n_rows, gn=None, relu=False, res=None, **kw):
    """Differentiable row block.  gn: nn.GroupNorm or None."""
    spec = BlockSpec(n_rows=n_rows, rels=rels, gn=gn is not None, relu=relu, has_res=res is not None,
                     eps=gn.eps if gn is not None else ops.EPS, **kw)
    gw, gb = (gn.weight, gn.bias) if gn is not None else (None, None)
    return RowBlockFn.apply(spec, len(srcs), len(weights), *srcs, *weights, gw, gb, res)


def linear_gn(x, weight, gn=None, relu=False, res=None, col0=0):
    """[ReLU]([GN](x W[:, col0:col0+128]^T) [+ res])."""
    return row_block([x], [weight], [Rel(0, 0, L.REL_IDENT, 0, col0)], x.shape[0], gn=gn, relu=relu, res=res)


def gn_act(x, gn=None, relu=False, res=None):
    gw, gb = (gn.weight, gn.bias) if gn is not None else (None, None)
    return GNActFn.apply(x, gw, gb, res, relu, gn.eps if gn is not None else ops.EPS)


class PredLossFn(torch.autograd.Function):
    """PredLoss's two sums (reference lanegcn.py:740-807) in one launch, gradients in one more (csrc/lgcn_loss.hip).
    Returns (cls_loss, reg_loss, counts [2] int32 on the device: num_cls, num_reg)."""

    @staticmethod
    def forward(ctx, cls, reg, gt, has, cfg):
        cls, reg = cls.contiguous(), reg.contiguous()
        sums, counts, sel = ops.pred_loss_fwd(cls, reg, gt, has, cfg)
        ctx.save_for_backward(cls, reg, gt, has, sel)
        ctx.cfg = cfg
        ctx.mark_non_differentiable(counts)
        return sums[0], sums[1], counts

    @staticmethod
    def backward(ctx, g_cls, g_reg, _):
        cls, reg, gt, has, sel = ctx.saved_tensors
        z = lambda g: torch.zeros(1, dtype=torch.float32, device=cls.device) if g is None else g.reshape(1).float().contiguous()
        dcls, dreg = ops.pred_loss_bwd(cls, reg, gt, has, ctx.cfg, sel, z(g_cls), z(g_reg))
        return dcls, dreg, None, None, None
