"""Graph construction on the GPU: the two producers of the lane relations the hot path
consumes, behind the reference's function names.

  preprocess(graph, cross_dist, cross_angle=None)   reference preprocess_data.py:287-392: left / right node adjacency of
      one scene from the lane-level left / right / pre / suc pairs (dense N x N distance + lane-pair mask + row argmin
      + 6 m and pi / 4 tests there; one wave per node here, lgcn_cross_edges).
  dilated_nbrs(nbr, num_nodes, num_scales)          reference data.py:520-534: A^(2^i) by repeated boolean squaring
      (lgcn_bool_square*), device tensors in and out.

Both need CUDA tensors (no CPU fallback).  `cross_angle`: the reference's optional sector test reads a module global
`config` (preprocess_data.py:310-312) that the module never defines -- `config` is a local of its main() (:44) -- so
passing it raises NameError there; it raises the same here, and the reference's own call (:250) never passes it.
"""
from typing import Dict

import numpy as np
import torch

from . import _lib as L
from . import ops


def dilated_nbrs(nbr: Dict, num_nodes: int, num_scales: int):
    u, v = torch.as_tensor(nbr["u"]), torch.as_tensor(nbr["v"])
    if not (u.is_cuda and v.is_cuda):
        raise L.LgcnError("dilated_nbrs: the HIP path needs CUDA tensors (host arrays: lanegcn_amd.data.dilated_nbrs)")
    return ops.dilated_nbrs(u.long(), v.long(), int(num_nodes), int(num_scales))


def preprocess(graph: Dict, cross_dist: float, cross_angle=None) -> Dict:
    """Same inputs and outputs as the reference: graph holds ctrs, feats [N,2], lane_idcs [N], pre_pairs, suc_pairs,
    left_pairs, right_pairs [k,2] (LongTensors on the GPU, as after to_long(gpu(.))) and idx; returns
    {"left": {"u", "v"}, "right": {"u", "v"}, "idx"} with int16 numpy index arrays."""
    if cross_angle is not None:        # as the reference: its branch dies on an undefined global (see the module docstring)
        raise NameError("name 'config' is not defined")
    lane_idcs = graph["lane_idcs"]
    if not (torch.is_tensor(lane_idcs) and lane_idcs.is_cuda):
        raise L.LgcnError("preprocess: the HIP path needs CUDA tensors (no CPU fallback)")
    num_lanes = int(lane_idcs[-1].item()) + 1                      # :292
    out = {}
    for side in ("left", "right"):
        pairs = graph[side + "_pairs"]
        if len(pairs) > 0:                                         # :317 / :355
            u, v = ops.cross_edges(graph["ctrs"], graph["feats"], lane_idcs, num_lanes, pairs, graph["pre_pairs"],
                                   graph["suc_pairs"], cross_dist)
            out[side] = {"u": u.cpu().numpy().astype(np.int16), "v": v.cpu().numpy().astype(np.int16)}
        else:
            out[side] = {"u": np.zeros(0, np.int16), "v": np.zeros(0, np.int16)}
    out["idx"] = graph["idx"]
    return out
