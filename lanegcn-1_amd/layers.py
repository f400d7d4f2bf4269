"""Building blocks with the reference's names and constructor arguments (layers.py).

``Linear`` (reference layers.py:65-87) is the only block the graph hot path uses; its CUDA
forward is one ``lgcn_agg_mlp`` launch.  ``Conv1d`` / ``Res1d`` / ``LinearRes`` serve ActorNet and
PredNet (outside the hot path) and run on stock PyTorch-ROCm ops.
"""
from math import gcd

import torch
from torch import nn
from torch.nn import functional as F

from . import _lib as L
from . import ops


def _norm(norm, ng, n_out, dims):
    if norm not in ("GN", "BN", "SyncBN"):
        raise AssertionError(norm)
    if norm == "GN":
        return nn.GroupNorm(gcd(ng, n_out), n_out)
    if norm == "BN":
        return nn.BatchNorm1d(n_out)
    raise SystemExit("SyncBN has not been added!")


def group_norm1(x, norm, relu=False, res=None):
    """out = [ReLU](norm(x) [+ res]).  Apply `norm` so that its backward is correct on this stack, and in one launch
    where no gradient is needed: [n, C, L] inputs under GroupNorm(1 group) run on lgcn_gn_cl (stock ATen: three
    launches for the norm plus one each for the residual add and the ReLU).

    Stock PyTorch-ROCm 2.10.0+rocm7.0 returns wrong dgamma / dbeta from GroupNorm's backward on the GPU once
    the batch dimension exceeds ~128 (2-D and 3-D inputs alike; forward and dx are right) -- measured with
    tools/check_aten_gn.py.  When gradients are needed on a CUDA tensor, GroupNorm(1 group) therefore runs on
    the HIP row kernels ([rows,128] inputs) or on an explicit mean/var formula made of basic ops."""
    if (not isinstance(norm, nn.GroupNorm) or norm.num_groups != 1 or not x.is_cuda
            or not ops.wants_grad(x, norm.weight, norm.bias, res)):
        if (isinstance(norm, nn.GroupNorm) and norm.num_groups == 1 and x.is_cuda and x.dim() == 3
                and x.dtype == torch.float32 and x.shape[1] * x.shape[2] <= 16384 and x.shape[0] > 0):
            return ops.gn_cl(x.contiguous(), norm.weight, norm.bias, norm.eps, res=None if res is None else res.contiguous(),
                             relu=relu)
        out = norm(x)
        if res is not None:
            out = out + res
        return F.relu(out) if relu else out
    if x.dim() == 3 and x.dtype == torch.float32 and 0 < x.shape[1] * x.shape[2] <= 16384 and x.shape[0] > 0:
        from . import autograd as A
        return A.gn_cl_act(x, norm, relu=relu, res=None if res is None else res.contiguous())
    if res is not None:
        out = group_norm1(x, norm) + res
        return F.relu(out) if relu else out
    if x.dim() == 2 and x.shape[1] == ops.C_FEAT:
        from . import autograd as A
        return A.gn_act(x.contiguous(), gn=norm, relu=relu)     # the ReLU is part of the Function (no in-place edit
    dims = tuple(range(1, x.dim()))                             # of its saved output afterwards)
    mean = x.mean(dims, keepdim=True)
    var = x.var(dims, unbiased=False, keepdim=True)
    shape = [1, -1] + [1] * (x.dim() - 2)
    out = (x - mean) * torch.rsqrt(var + norm.eps) * norm.weight.view(shape) + norm.bias.view(shape)
    return F.relu(out) if relu else out


class Linear(nn.Module):
    """no-bias Linear -> GroupNorm -> optional ReLU (reference layers.py:65-87)."""

    def __init__(self, n_in, n_out, norm="GN", ng=32, act=True):
        super().__init__()
        self.linear = nn.Linear(n_in, n_out, bias=False)
        self.norm = _norm(norm, ng, n_out, 1)
        self.relu = nn.ReLU(inplace=True)
        self.act = act

    def _hot_shaped(self, x):
        return (x.dim() == 2 and self.linear.out_features == ops.C_FEAT
                and self.linear.in_features == ops.C_FEAT and isinstance(self.norm, nn.GroupNorm)
                and self.norm.num_groups == 1)

    def forward(self, x):
        if self._hot_shaped(x):
            # the graph hot path's shape: HIP only (CUDA tensors, no CPU fallback)
            if ops.wants_grad(x, *ops.module_params(self)):
                from . import autograd as A
                return A.linear_gn(x, self.linear.weight, gn=self.norm, relu=self.act)
            flags = L.F_GN1 | (L.F_RELU1 if self.act else 0)
            return ops.agg_mlp(x.shape[0], [ops.RelSpec(x, ops.packed(self.linear.weight))], flags,
                               gn1=(self.norm.weight, self.norm.bias), eps=self.norm.eps)
        # other shapes (AttDest 256->128 in PredNet) are outside the hot path: stock ATen ops
        return group_norm1(self.linear(x), self.norm, relu=self.act)


class Conv1d(nn.Module):
    """Conv1d(bias=False) -> norm -> optional ReLU (reference layers.py:40-62); ActorNet only."""

    def __init__(self, n_in, n_out, kernel_size=3, stride=1, norm="GN", ng=32, act=True):
        super().__init__()
        self.conv = nn.Conv1d(n_in, n_out, kernel_size=kernel_size, padding=(int(kernel_size) - 1) // 2,
                              stride=stride, bias=False)
        self.norm = _norm(norm, ng, n_out, 1)
        self.relu = nn.ReLU(inplace=True)
        self.act = act

    def forward(self, x):
        return group_norm1(self.conv(x), self.norm, relu=self.act)


class Res1d(nn.Module):
    """Two-conv residual block over the time axis (reference layers.py:142-190); ActorNet only."""

    def __init__(self, n_in, n_out, kernel_size=3, stride=1, norm="GN", ng=32, act=True):
        super().__init__()
        pad = (int(kernel_size) - 1) // 2
        self.conv1 = nn.Conv1d(n_in, n_out, kernel_size=kernel_size, stride=stride, padding=pad, bias=False)
        self.conv2 = nn.Conv1d(n_out, n_out, kernel_size=kernel_size, padding=pad, bias=False)
        self.relu = nn.ReLU(inplace=True)
        self.bn1 = _norm(norm, ng, n_out, 1)
        self.bn2 = _norm(norm, ng, n_out, 1)
        if stride != 1 or n_out != n_in:
            self.downsample = nn.Sequential(
                nn.Conv1d(n_in, n_out, kernel_size=1, stride=stride, bias=False), _norm(norm, ng, n_out, 1))
        else:
            self.downsample = None
        self.act = act

    def forward(self, x):
        out = group_norm1(self.conv1(x), self.bn1, relu=True)
        if self.downsample is not None:
            x = group_norm1(self.downsample[0](x), self.downsample[1])
        return group_norm1(self.conv2(out), self.bn2, relu=self.act, res=x)     # norm + residual + ReLU: one launch


class LinearRes(nn.Module):
    """Two-layer residual MLP (reference layers.py:193-238); PredNet only."""

    def __init__(self, n_in, n_out, norm="GN", ng=32):
        super().__init__()
        self.linear1 = nn.Linear(n_in, n_out, bias=False)
        self.linear2 = nn.Linear(n_out, n_out, bias=False)
        self.relu = nn.ReLU(inplace=True)
        self.norm1 = _norm(norm, ng, n_out, 1)
        self.norm2 = _norm(norm, ng, n_out, 1)
        if n_in != n_out:
            self.transform = nn.Sequential(nn.Linear(n_in, n_out, bias=False), _norm(norm, ng, n_out, 1))
        else:
            self.transform = None

    def _hot_shaped(self, x):
        return (x.is_cuda and x.dim() == 2 and self.transform is None and self.linear1.in_features == ops.C_FEAT
                and self.linear1.out_features == ops.C_FEAT and isinstance(self.norm1, nn.GroupNorm)
                and self.norm1.num_groups == 1)

    def block_kw(self, x):
        """Keywords of the inference row block (ops.agg_mlp / one half of ops.agg_mlp_pair) for a contiguous x."""
        full = L.F_GN1 | L.F_RELU1 | L.F_GEMM2 | L.F_GN2 | L.F_RES | L.F_RELU2
        return dict(n_rows=x.shape[0], rels=[ops.RelSpec(x, ops.packed(self.linear1.weight))], flags=full,
                    gn1=(self.norm1.weight, self.norm1.bias), wp2=ops.packed(self.linear2.weight),
                    gn2=(self.norm2.weight, self.norm2.bias), res=x, eps=self.norm1.eps)

    def forward(self, x):
        if self._hot_shaped(x):
            # 128 -> 128: the fused two-stage row block of the hot path (one launch; PredNet heads, lanegcn.py:587-600)
            from . import autograd as A
            x = x.contiguous()
            if ops.wants_grad(x, *ops.module_params(self)):
                spec = A.BlockSpec(n_rows=x.shape[0], rels=[A.Rel(0, 0, L.REL_IDENT)], gn=True, relu=True, has_res=True,
                                   eps=self.norm1.eps)
                return A.LaneConvFn.apply(spec, x, self.norm1.weight, self.norm1.bias, self.linear2.weight,
                                          self.norm2.weight, self.norm2.bias, self.linear1.weight)
            return ops.agg_mlp(**self.block_kw(x))
        out = group_norm1(self.linear1(x), self.norm1, relu=True)
        out = group_norm1(self.linear2(out), self.norm2)
        if self.transform is not None:
            x = group_norm1(self.transform[0](x), self.transform[1])
        out = out + x
        return F.relu(out)


class Null(nn.Module):
    def forward(self, x):
        return x
