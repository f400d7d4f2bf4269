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
            ops._no_grad_guard(x, *self.parameters())
            flags = L.F_GN1 | (L.F_RELU1 if self.act else 0)
            return ops.agg_mlp(x.shape[0], [ops.RelSpec(x, ops.packed(self.linear.weight))], flags,
                               gn1=(self.norm.weight, self.norm.bias), eps=self.norm.eps)
        # other shapes (AttDest 256->128 in PredNet) are outside the hot path: stock ATen ops
        out = self.norm(self.linear(x))
        return self.relu(out) if self.act else out


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
        out = self.norm(self.conv(x))
        return self.relu(out) if self.act else out


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
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        out = out + (x if self.downsample is None else self.downsample(x))
        return self.relu(out) if self.act else out


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

    def forward(self, x):
        out = self.relu(self.norm1(self.linear1(x)))
        out = self.norm2(self.linear2(out))
        out = out + (x if self.transform is None else self.transform(x))
        return self.relu(out)


class Null(nn.Module):
    def forward(self, x):
        return x
