"""Flat scene store: the on-disk counterpart of engine.FlatBatch.

The reference keeps preprocessed scenes as a pickle of nested dicts with int16 index arrays
(preprocess_data.py:230-263) and rebuilds batches from ~40 small arrays per scene.  Here a split is ONE
`.npz` of concatenated arrays plus per-scene offset tables; a batch is cut out with a handful of slices and
goes to the device as one copy per array.  Everything is plain numpy (loadable with allow_pickle=False).

Layout (S scenes, R = 2 * num_scales + 2 relations in the order pre0, suc0, ..., left, right):
  node_off [S+1], actor_off [S+1]                      int64 prefix sums of nodes / actors per scene
  node_ctrs, node_feats, turn [N,2]; control, intersect [N]     float32
  actor_ctrs [A,2], actor_feats [A,20,3], rot [S,2,2], orig [S,2]   float32
  gt_preds [A,30,2] float32, has_preds [A,30] bool
  edge_off [R, S+1] int64: edges of relation r of scene s are  u[r][edge_off[r,s]:edge_off[r,s+1]]
  edge_u, edge_v [sum E] int32 (scene-local node indices), stored relation-major: rel_off [R+1]
"""
from typing import Dict, List, Sequence

import numpy as np
import torch

from .engine import FlatBatch

_NODE = ("ctrs", "feats", "turn", "control", "intersect")


def _npy(x):
    return x.numpy() if torch.is_tensor(x) else np.asarray(x)


def _relations(graph) -> List[Dict]:
    rels = []
    for i in range(len(graph["pre"])):
        rels += [graph["pre"][i], graph["suc"][i]]
    return rels + [graph["left"], graph["right"]]


def write_scenes(path: str, scenes: Sequence[Dict]) -> None:
    """Write scene dicts (reference schema) as one flat store."""
    S = len(scenes)
    graphs = [s["graph"] for s in scenes]
    R = 2 * len(graphs[0]["pre"]) + 2
    node_off = np.zeros(S + 1, np.int64)
    np.cumsum([int(g["num_nodes"]) for g in graphs], out=node_off[1:])
    actor_off = np.zeros(S + 1, np.int64)
    np.cumsum([len(s["ctrs"]) for s in scenes], out=actor_off[1:])
    edge_off = np.zeros((R, S + 1), np.int64)
    us, vs = [[] for _ in range(R)], [[] for _ in range(R)]
    for j, g in enumerate(graphs):
        for r, rel in enumerate(_relations(g)):
            u, v = _npy(rel["u"]).reshape(-1), _npy(rel["v"]).reshape(-1)
            us[r].append(u.astype(np.int32))
            vs[r].append(v.astype(np.int32))
            edge_off[r, j + 1] = edge_off[r, j] + len(u)
    rel_off = np.zeros(R + 1, np.int64)
    np.cumsum(edge_off[:, -1], out=rel_off[1:])
    cat = lambda parts, dt: np.concatenate(parts).astype(dt) if parts else np.zeros(0, dt)
    out = dict(
        num_scales=np.int64(len(graphs[0]["pre"])), node_off=node_off, actor_off=actor_off, edge_off=edge_off,
        rel_off=rel_off,
        edge_u=cat([x for r in range(R) for x in us[r]], np.int32), edge_v=cat([x for r in range(R) for x in vs[r]], np.int32),
        actor_ctrs=cat([_npy(s["ctrs"]) for s in scenes], np.float32),
        actor_feats=cat([_npy(s["feats"]) for s in scenes], np.float32),
        rot=np.stack([_npy(s["rot"]) for s in scenes]).astype(np.float32),
        orig=np.stack([_npy(s["orig"]) for s in scenes]).astype(np.float32),
    )
    for k in _NODE:
        out["node_" + k if k in ("ctrs", "feats") else k] = cat([_npy(g[k]) for g in graphs], np.float32)
    if "gt_preds" in scenes[0]:
        out["gt_preds"] = cat([_npy(s["gt_preds"]) for s in scenes], np.float32)
        out["has_preds"] = cat([_npy(s["has_preds"]) for s in scenes], bool)
    np.savez(path, **out)


class FlatSceneFile:
    """Read side: `batch(indices)` cuts a FlatBatch (+ actor tracks) for any list of scene indices."""

    def __init__(self, path: str):
        with np.load(path, allow_pickle=False) as z:
            self.a = {k: z[k] for k in z.files}
        self.n_scenes = len(self.a["node_off"]) - 1
        self.num_scales = int(self.a["num_scales"])

    def __len__(self):
        return self.n_scenes

    def _rows(self, off, idx):
        return np.concatenate([np.arange(off[i], off[i + 1]) for i in idx]) if len(idx) else np.zeros(0, np.int64)

    def batch(self, indices: Sequence[int], device=None):
        """-> (FlatBatch, extras) with extras = {actor_feats [A,3,20], rot [A,2,2], orig [A,2], sizes,
        gt_preds, has_preds}: everything FullNetEngine / HotPathEngine need, no per-scene dicts."""
        a = self.a
        idx = list(indices)
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        up = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
        n_nodes = np.array([a["node_off"][i + 1] - a["node_off"][i] for i in idx], np.int64)
        n_act = np.array([a["actor_off"][i + 1] - a["actor_off"][i] for i in idx], np.int64)
        node_off = np.zeros(len(idx) + 1, np.int64)
        np.cumsum(n_nodes, out=node_off[1:])
        actor_off = np.zeros(len(idx) + 1, np.int64)
        np.cumsum(n_act, out=actor_off[1:])
        nrow, arow = self._rows(a["node_off"], idx), self._rows(a["actor_off"], idx)
        R = a["edge_off"].shape[0]
        pieces, seg_len, seg_base, rel_slices, n_edges = [], [], [], [], []
        pos = 0
        for r in range(R):
            spans = []
            for which in ("edge_u", "edge_v"):
                begin = pos
                for j, i in enumerate(idx):
                    lo, hi = a["rel_off"][r] + a["edge_off"][r, i], a["rel_off"][r] + a["edge_off"][r, i + 1]
                    pieces.append(a[which][lo:hi].astype(np.int64))
                    seg_len.append(hi - lo)
                    seg_base.append(node_off[j])
                    pos += hi - lo
                spans.append((int(begin), int(pos)))
            rel_slices.append((spans[0], spans[1]))
            n_edges.append(spans[0][1] - spans[0][0])
        seg_off = np.zeros(len(seg_len) + 1, np.int64)
        np.cumsum(seg_len, out=seg_off[1:])
        fb = FlatBatch(
            n_scenes=len(idx), n_nodes=int(node_off[-1]), n_actors=int(actor_off[-1]), num_scales=self.num_scales,
            node_ctrs=up(a["node_ctrs"][nrow]), node_feats=up(a["node_feats"][nrow]), turn=up(a["turn"][nrow]),
            control=up(a["control"][nrow]), intersect=up(a["intersect"][nrow]), actor_ctrs=up(a["actor_ctrs"][arow]),
            node_off=up(node_off.astype(np.int32)), actor_off=up(actor_off.astype(np.int32)),
            idx_local=up(np.concatenate(pieces) if pieces else np.zeros(0, np.int64)), seg_off=up(seg_off),
            seg_base=up(np.asarray(seg_base, np.int64)), rel_slices=rel_slices,
            cap_a2m=int(np.dot(n_nodes, n_act)), cap_a2a=int(np.dot(n_act, n_act)), n_edges=n_edges)
        rep = np.repeat(np.asarray(idx), n_act)
        extras = {"actor_feats": up(a["actor_feats"][arow].transpose(0, 2, 1)), "rot": up(a["rot"][rep]),
                  "orig": up(a["orig"][rep]), "sizes": [int(x) for x in n_act]}
        if "gt_preds" in a:
            extras["gt_preds"], extras["has_preds"] = up(a["gt_preds"][arow]), up(a["has_preds"][arow])
        return fb, extras
