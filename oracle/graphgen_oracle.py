"""TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline may import it; the product may
not).  CPU restatement, in numpy, of the reference's graph construction:

  preprocess       reference preprocess_data.py:287-392 (cross_angle = None): left / right node adjacency of a scene
  dilated_nbrs     reference data.py:520-534: see oracle/lanegcn_oracle.dilated_nbrs (scipy restatement, pinned to the
                   reference's own output by tests/test_oracle_golden.py)

Pinned by tests/golden/graphgen_b6.npz: inputs and outputs of the reference's own `preprocess` run in this container
(tests/golden/make_golden.py::graphgen_fixture).
"""
import numpy as np


def _side(ctrs, feats, lane_idcs, num_lanes, side_pairs, pre_pairs, suc_pairs, cross_dist):
    n = len(lane_idcs)
    if len(side_pairs) == 0:                                               # :317 / :355
        return np.zeros(0, np.int16), np.zeros(0, np.int16)
    ctrs = ctrs.astype(np.float32)
    d = ctrs[:, None, :] - ctrs[None, :, :]                                # :294
    dist = np.sqrt(((d * d)[..., 0] + (d * d)[..., 1]).astype(np.float32)).astype(np.float32)     # :295 (fp32 throughout)
    pre = np.zeros((num_lanes, num_lanes), np.float32)                     # :310-313
    pre[pre_pairs[:, 0], pre_pairs[:, 1]] = 1
    suc = np.zeros((num_lanes, num_lanes), np.float32)
    suc[suc_pairs[:, 0], suc_pairs[:, 1]] = 1
    mat = np.zeros((num_lanes, num_lanes), np.float32)                     # :318-320
    mat[side_pairs[:, 0], side_pairs[:, 1]] = 1
    mat = (mat @ pre + mat @ suc + mat) > 0.5
    allowed = mat[lane_idcs[:, None], lane_idcs[None, :]]                  # :323
    dd = np.where(allowed, dist, np.float32(1e6))                          # :322-324
    min_idcs = dd.argmin(1)                                                # :328 (first of equals, as a CPU min(1))
    min_dist = dd[np.arange(n), min_idcs]
    keep = min_dist < np.float32(cross_dist)                               # :329
    ui, vi = np.arange(n)[keep], min_idcs[keep]
    f1, f2 = feats[ui].astype(np.float32), feats[vi].astype(np.float32)    # :332-337
    t1 = np.arctan2(f1[:, 1], f1[:, 0]).astype(np.float32)
    t2 = np.arctan2(f2[:, 1], f2[:, 0]).astype(np.float32)
    dt = np.abs(t1 - t2).astype(np.float32)
    m = dt > np.float32(np.pi)                                             # :338-339
    dt[m] = np.abs(dt[m] - np.float32(2 * np.pi))
    m = dt < np.float32(0.25 * np.pi)                                      # :340
    return ui[m].astype(np.int16), vi[m].astype(np.int16)


def heading_margin(feats, ui, vi):
    """Distance of every kept / dropped candidate's heading difference from the pi / 4 threshold (tests use it to make
    sure a fixture does not hinge on the last bit of atan2)."""
    f1, f2 = feats[ui].astype(np.float64), feats[vi].astype(np.float64)
    dt = np.abs(np.arctan2(f1[:, 1], f1[:, 0]) - np.arctan2(f2[:, 1], f2[:, 0]))
    dt = np.where(dt > np.pi, np.abs(dt - 2 * np.pi), dt)
    return np.abs(dt - 0.25 * np.pi)


def preprocess(graph, cross_dist):
    """graph: numpy arrays ctrs, feats [N,2], lane_idcs [N], pre_pairs, suc_pairs, left_pairs, right_pairs [k,2]."""
    lane_idcs = np.asarray(graph["lane_idcs"]).astype(np.int64)
    num_lanes = int(lane_idcs[-1]) + 1                                     # :292
    out = {}
    for side in ("left", "right"):
        u, v = _side(np.asarray(graph["ctrs"]), np.asarray(graph["feats"]), lane_idcs, num_lanes,
                     np.asarray(graph[side + "_pairs"]).reshape(-1, 2).astype(np.int64),
                     np.asarray(graph["pre_pairs"]).reshape(-1, 2).astype(np.int64),
                     np.asarray(graph["suc_pairs"]).reshape(-1, 2).astype(np.int64), cross_dist)
        out[side] = {"u": u, "v": v}
    return out
