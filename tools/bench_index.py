"""Micro-benchmark of the index stage on a synthetic S2 batch: lgcn_index_build with and without the pair searches,
the pair searches alone, the separate entry points.  Usage: python tools/bench_index.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lanegcn_amd  # noqa: E402,F401
from lanegcn_amd import data as gen  # noqa: E402
from lanegcn_amd import lanegcn as M  # noqa: E402
from lanegcn_amd import ops  # noqa: E402
from lanegcn_amd.engine import collate_flat  # noqa: E402
from tools.bench_agg import timeit  # noqa: E402


def main():
    fb = collate_flat(gen.synth_batch("S2", seed=100, n_scenes=32))
    cfg = M.config
    searches = ((fb.node_ctrs, fb.node_off, fb.actor_ctrs, fb.actor_off, cfg["actor2map_dist"], fb.cap_a2m),
                (fb.actor_ctrs, fb.actor_off, fb.node_ctrs, fb.node_off, cfg["map2actor_dist"], fb.cap_a2m),
                (fb.actor_ctrs, fb.actor_off, fb.actor_ctrs, fb.actor_off, cfg["actor2actor_dist"], fb.cap_a2a))
    dev = fb.node_ctrs.device
    cnt = ops.index_counters(fb.n_nodes, len(fb.rel_slices), dev)
    bufs = [ops.pairs_alloc(s[0].shape[0], fb.n_scenes, s[5], dev) for s in searches]

    def separate():
        ops.pairs_build_multi(searches, True, bufs=bufs)
        g64, _ = ops.graph_gather_indices(fb.idx_local, fb.seg_off, fb.seg_base)
        ops.csr_build([g64[a:b] for (a, b), _ in fb.rel_slices], [g64[a:b] for _, (a, b) in fb.rel_slices], fb.n_nodes)

    with torch.no_grad():
        print("index_build, 3 pair searches : %.2f us" % timeit(lambda: ops.index_build(
            fb.idx_local, fb.seg_off, fb.seg_base, fb.rel_slices, fb.n_nodes, searches, True, bufs=bufs, cnt=cnt)))
        print("index_build, no pair search  : %.2f us" % timeit(lambda: ops.index_build(
            fb.idx_local, fb.seg_off, fb.seg_base, fb.rel_slices, fb.n_nodes, (), True, cnt=cnt)))
        for k, name in enumerate(("a2m", "m2a", "a2a")):
            print("index_build, %s search only : %.2f us" % (name, timeit(lambda: ops.index_build(
                fb.idx_local, fb.seg_off, fb.seg_base, fb.rel_slices, fb.n_nodes, searches[k:k + 1], True, bufs=bufs[k:k + 1], cnt=cnt))))
        print("pairs_build_multi alone      : %.2f us" % timeit(lambda: ops.pairs_build_multi(searches, True, bufs=bufs)))
        print("separate entry points        : %.2f us" % timeit(separate))


if __name__ == "__main__":
    main()
