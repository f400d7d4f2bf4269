"""Distinct source rows per LaneConv work item of a synthetic batch, per row-block shape (how often a plan has to
split a row block's units into several items for a given LDS capacity).  Usage: python tools/lc_plan_stats.py [scenes]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lanegcn_amd  # noqa: E402,F401
from lanegcn_amd import data as gen  # noqa: E402
from lanegcn_amd import ops  # noqa: E402
from lanegcn_amd.engine import collate_flat  # noqa: E402


def main():
    scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    ops.set_mma("f16x2")
    fb = collate_flat(gen.synth_batch("S2", seed=100, n_scenes=scenes))
    g64, _ = ops.graph_gather_indices(fb.idx_local, fb.seg_off, fb.seg_base)
    plan = ops.csr_build([g64[a:b] for (a, b), _ in fb.rel_slices], [g64[a:b] for _, (a, b) in fb.rel_slices], fb.n_nodes)
    for variant in (0, 1, 2):
        M, cap_max = ops.lc_config(variant=variant)
        for cap in sorted({cap_max, M + 64, M + 104}):
            if cap > cap_max:
                continue
            lcp = ops.lc_plan(plan, n_groups=1, cap=cap, variant=variant)
            n_blocks = (fb.n_nodes + M - 1) // M
            hdr = lcp.plan[: n_blocks * 15 * 8].cpu().numpy().reshape(n_blocks * 15, 8)
            live = hdr[hdr[:, 0] > 0]
            per_block = np.bincount(np.repeat(np.arange(n_blocks), 15)[hdr[:, 0] > 0], minlength=n_blocks)
            print("M=%3d cap=%3d: items per row block mean %.2f max %d; sources per item mean %.0f max %d" % (
                M, cap, per_block.mean(), per_block.max(), live[:, 1].mean(), live[:, 1].max()))


if __name__ == "__main__":
    main()
