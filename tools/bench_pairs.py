"""Micro-benchmark of the pair MLP kernel (lgcn_att_pairs_ws) on the A2A / M2A / A2M pair sets of a synthetic S2 batch.
Usage: python tools/bench_pairs.py [--lib stamps]   (--lib stamps: the diagnostic library, for its LGCN_EXP_* knobs)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lanegcn_amd  # noqa: E402,F401
from lanegcn_amd import _lib as L  # noqa: E402

if "--lib" in sys.argv:
    L.LIB_PATH = os.path.join(os.path.dirname(L.LIB_PATH), "liblgcn_%s.so" % sys.argv[sys.argv.index("--lib") + 1])
from lanegcn_amd import data as gen  # noqa: E402
from lanegcn_amd import lanegcn as M  # noqa: E402
from lanegcn_amd import ops  # noqa: E402
from lanegcn_amd.engine import collate_flat  # noqa: E402
from tools.bench_agg import timeit  # noqa: E402


def main():
    ops.set_mma("f16x2")
    torch.manual_seed(0)
    att = M.Att(128, 128).cuda().eval()
    fb = collate_flat(gen.synth_batch("S2", seed=100, n_scenes=32))
    cfg = M.config
    sets = {"a2m": (fb.node_ctrs, fb.node_off, fb.actor_ctrs, fb.actor_off, cfg["actor2map_dist"], fb.cap_a2m),
            "m2a": (fb.actor_ctrs, fb.actor_off, fb.node_ctrs, fb.node_off, cfg["map2actor_dist"], fb.cap_a2m),
            "a2a": (fb.actor_ctrs, fb.actor_off, fb.actor_ctrs, fb.actor_off, cfg["actor2actor_dist"], fb.cap_a2a)}
    with torch.no_grad():
        for name, s in sets.items():
            ps = ops.pairs_build(*s)
            T, S = s[0].shape[0], s[2].shape[0]
            U, V = torch.randn(T, 128, device="cuda"), torch.randn(S, 128, device="cuda")
            c0 = att.ctx[0]
            m = torch.empty((ps.cap, 128), device="cuda")
            args = (ps, att.dist[0].weight, att.dist[0].bias, ops.packed(att.dist[2].linear.weight), M._gn(att.dist[2].norm),
                    ops.packed(c0.linear.weight, 0, 128), U, V, M._gn(c0.norm))
            seg = 0 if name == "a2m" else 16
            print("%s: P = %d, seg = %d: %.2f us" % (name, ps.count(), seg, timeit(lambda: ops.att_pairs(*args, m=m, seg=seg))))


if __name__ == "__main__":
    main()
