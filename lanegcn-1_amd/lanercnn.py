"""Second consumer of the two hot kernels: the graph modules of the reference's fork
model lanercnn.py on the same HIP row-block / LaneConv / pair kernels.

Same class names, constructor arguments, forward signatures and state_dict names as the reference:
  LaneInput      lanercnn.py:280-351  map_fc(8 -> 128) + index_add_ of agt_fc(80 -> 128) over a2m edges -> GN -> ReLU
  LaneRoI        lanercnn.py:354-430  Linear(input_dim -> 128, GN, ReLU) + 4 LaneConv layers
  GlobalGraphNet lanercnn.py:517-600  4 LaneConv layers on a given feature
  LanePooling    lanercnn.py:433-514  distance-gated pooling between two lane graphs: the Att pattern with a 4-d
                                      relative pose instead of the 2-d offset and without a query term

Inference (no_grad) runs on the HIP kernels; under autograd LaneRoI / GlobalGraphNet train through the LaneConv
autograd path of lanegcn.py, LanePooling and LaneInput through the row-block / pair / gather Functions of autograd.py
(the same composition as Att.run_train): every 128-d contraction of the backward is a HIP launch as well.
"""
from math import gcd
from typing import Dict, List

import numpy as np
import torch
from torch import Tensor, nn
from torch.nn import functional as F

from . import _lib as L
from . import autograd as A
from . import ops
from .lanegcn import _fuse_modules, _gn, build_pairs, lane_conv, lane_conv_train, lane_plan, lane_plan_t
from .layers import Linear


def _need_cuda(*ts):
    for t in ts:
        if torch.is_tensor(t) and not t.is_cuda:
            raise L.LgcnError("lanercnn modules need CUDA tensors (the HIP hot path has no CPU fallback)")


class LaneInput(nn.Module):
    """Lane-RoI input encoder (reference lanercnn.py:280-351)."""

    def __init__(self, config):
        super().__init__()
        map_dim = config["n_map"]
        self.map_fc = nn.Linear(8, map_dim, bias=False)
        self.agt_fc = nn.Linear(80, map_dim, bias=False)
        self.bn = nn.GroupNorm(gcd(1, map_dim), map_dim)
        self.relu = nn.ReLU(inplace=True)

    def forward(self, graph: Dict) -> Tensor:
        map_feats = torch.cat(graph["feats"], 0)            # [nodes, 8]
        agt_feats = torch.cat(graph["agent_feat"], 0)       # [agts, 80]
        _need_cuda(map_feats, agt_feats)
        train = ops.wants_grad(map_feats, agt_feats, *ops.module_params(self))
        n = map_feats.shape[0]
        # the two Linears have K = 8 / 80 (not 128-d contractions): stock ops; agt_fc commutes with the gather
        base = self.map_fc(map_feats)
        agt = self.agt_fc(agt_feats)
        u, v = graph["a2m"]["u"].long(), graph["a2m"]["v"].long()
        if u.numel() > 0:
            # one relation: key(n, 0) = n, i.e. a plain CSR by node (sized for the larger index space: the builder
            # bounds-checks sources and destinations against the same count)
            rows = max(n, agt.shape[0])
            plan = ops.csr_build([v], [u], rows)
            if train:      # the gather's transpose = the same gather over the CSR by source
                base = base + A.GatherSumFn.apply(agt, plan, ops.csr_build([u], [v], rows), n)
            else:
                base = base + ops.gather_sum(agt, plan.rowptr, plan.col, n)
        if train:
            return A.gn_act(base.contiguous(), gn=self.bn, relu=True)
        return ops.gn_fwd(base.contiguous(), _gn(self.bn), relu=True, eps=self.bn.eps)


class LaneRoI(nn.Module):
    """Lane-RoI encoder: input Linear + 4 LaneConv layers (reference lanercnn.py:354-430)."""

    def __init__(self, config, input_dim):
        super().__init__()
        self.config = config
        map_dim = config["n_map"]
        self.input = Linear(input_dim, map_dim, norm="GN", ng=1, act=True)
        self.fuse = _fuse_modules(map_dim, config["num_scales"])
        self.relu = nn.ReLU(inplace=True)

    def forward(self, feat: Tensor, graph: Dict) -> Tensor:
        _need_cuda(feat)
        feat = self.input(feat)
        if ops.wants_grad(feat, *ops.module_params(self)):
            return lane_conv_train(self.fuse, feat, lane_plan(graph), lane_plan_t(graph), len(graph["pre"]))
        return ops.guarded(lambda: lane_conv(self.fuse, feat, lane_plan(graph), len(graph["pre"])))


class GlobalGraphNet(nn.Module):
    """4 LaneConv layers over the global lane graph (reference lanercnn.py:517-600)."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.fuse = _fuse_modules(config["n_map"], config["num_scales"])
        self.relu = nn.ReLU(inplace=True)

    def forward(self, feat: Tensor, graph: Dict):
        if len(graph["feats"]) == 0 or len(graph["pre"][-1]["u"]) == 0 or len(graph["suc"][-1]["u"]) == 0:
            temp = graph["feats"]                            # the reference returns a 1-tuple here (:538-544)
            return (temp.new().resize_(0),)
        _need_cuda(feat)
        if ops.wants_grad(feat, *ops.module_params(self)):
            return lane_conv_train(self.fuse, feat, lane_plan(graph), lane_plan_t(graph), len(graph["pre"]))
        return ops.guarded(lambda: lane_conv(self.fuse, feat, lane_plan(graph), len(graph["pre"])))


class LanePooling(nn.Module):
    """Distance-gated pooling of a context lane graph into a target lane graph (reference lanercnn.py:433-514)."""
    legacy_offsets = True     # scenes without a pair do not advance the index offsets (lanercnn.py:476-483)

    def __init__(self, in_dim: int, out_dim: int) -> None:
        super().__init__()
        in_dim, mid_dim, out_dim = 128, 128, 128             # the reference overrides its arguments (:438)
        self.input = nn.Linear(in_dim, mid_dim, bias=False)
        self.relpose = nn.Sequential(nn.Linear(4, in_dim), nn.ReLU(inplace=True))
        self.ctx = nn.Sequential(Linear(in_dim * 2, mid_dim, norm="GN", ng=1), nn.Linear(mid_dim, mid_dim, bias=False))
        self.mlp = nn.Sequential(Linear(mid_dim, mid_dim, norm="GN", ng=1),
                                 Linear(mid_dim, out_dim, norm="GN", ng=1, act=False))
        self.norm = nn.GroupNorm(gcd(1, 128), 128)
        self.relu = nn.ReLU(inplace=True)

    def forward(self, context_feat: Tensor, context_graph: Dict, target_feat: Tensor, target_graph: Dict,
                dist_th: float = 6.0, g2r: bool = False) -> Tensor:
        _need_cuda(context_feat, target_feat)
        if ops.wants_grad(context_feat, target_feat, *ops.module_params(self)):
            return self._run(context_feat, context_graph, target_feat, target_graph, dist_th, train=True)
        return ops.guarded(lambda: self._run(context_feat, context_graph, target_feat, target_graph, dist_th))

    def _run(self, context_feat, context_graph, target_feat, target_graph, dist_th, train=False):
        c_ctrs, t_ctrs = context_graph["ctrs"], target_graph["ctrs"]
        # The reference lists the pairs context-major (hi = context row, wi = target row) and index_add_s them by
        # TARGET (:509): for one target the contributions arrive in ascending context order.  Searching with the
        # target as the row side gives exactly those segments, contiguous and in that order (and the same numbering
        # quirk: a scene without pairs advances neither offset).
        idc = lambda ctrs: [torch.arange(len(c)) for c in ctrs]
        ps = build_pairs(idc(t_ctrs), t_ctrs, idc(c_ctrs), c_ctrs, dist_th, self.legacy_offsets)
        P = ps.count()
        if P == 0:
            raise RuntimeError("torch.cat(): expected a non-empty list of Tensors")          # lanercnn.py:484
        T = target_feat.shape[0]
        t_idx, c_idx = ps.hi[:P].long(), ps.wi[:P].long()
        c_pose = torch.cat(context_graph["pose"], 0)
        t_pose = torch.cat(target_graph["pose"], 0)
        h = F.relu(self.relpose[0](c_pose[c_idx] - t_pose[t_idx]))                        # [P,128]; K = 4: stock op
        w0 = self.ctx[0].linear.weight                                                     # [128, 256] = [feat | pose]
        if train:      # the differentiable composition of the same arithmetic (cf. Att.run_train)
            m0, m1 = self.mlp[0], self.mlp[1]
            per_ctx = A.linear_gn(context_feat, w0, col0=0)
            per_pair = A.linear_gn(h.contiguous(), w0, col0=128)
            no_query = torch.zeros((T, ops.C_FEAT), dtype=torch.float32, device=per_pair.device)
            pre = A.PairAddFn.apply(per_pair, no_query, per_ctx, ps)                    # + per_ctx[context row of the pair]
            m = A.gn_act(pre, gn=self.ctx[0].norm, relu=True)
            y = A.row_block([target_feat, m], [self.input.weight, self.ctx[1].weight],
                            [A.Rel(0, 0, L.REL_IDENT), A.Rel(1, 1, L.REL_RANGE)], T, gn=self.norm, relu=True,
                            rowptr=ps.rowptr, seg_ids=ps.hi, n_seg_rows=ps.n_pairs)
            y = A.linear_gn(y, m0.linear.weight, gn=m0.norm, relu=True)
            return A.linear_gn(y, m1.linear.weight, gn=m1.norm, relu=True, res=target_feat)
        per_ctx = ops.agg_mlp(context_feat.shape[0], [ops.RelSpec(context_feat, ops.packed(w0, 0, 128))], 0)
        per_pair = ops.agg_mlp(P, [ops.RelSpec(h.contiguous(), ops.packed(w0, 128, 128))], 0)
        zero_row = torch.zeros((1, ops.C_FEAT), dtype=torch.float32, device=per_pair.device)
        zero_idx = torch.zeros(P, dtype=torch.int32, device=per_pair.device)
        pre = ops.pair_add(per_pair, per_ctx, ps.wi, zero_row, zero_idx, ps.n_pairs, P)
        m = ops.gn_fwd(pre, _gn(self.ctx[0].norm), relu=True, eps=self.ctx[0].norm.eps)
        m0, m1 = self.mlp[0], self.mlp[1]
        # ctx.1 is linear: applied to the per-target segment sum (pairs sorted by target: a RANGE relation)
        y = ops.agg_mlp(T, [ops.RelSpec(target_feat, ops.packed(self.input.weight)),
                            ops.RelSpec(m, ops.packed(self.ctx[1].weight), L.REL_RANGE)],
                        L.F_GN1 | L.F_RELU1 | L.F_GEMM2 | L.F_GN2 | L.F_RELU2, rowptr=ps.rowptr, gn1=_gn(self.norm),
                        wp2=ops.packed(m0.linear.weight), gn2=_gn(m0.norm), eps=self.norm.eps)
        return ops.agg_mlp(T, [ops.RelSpec(y, ops.packed(m1.linear.weight))], L.F_GN1 | L.F_RES | L.F_RELU1,
                           gn1=_gn(m1.norm), res=target_feat, eps=m1.norm.eps)
