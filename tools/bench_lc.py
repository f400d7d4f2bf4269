"""Micro-benchmark of one LaneConv layer on a synthetic batch: the weight-stationary pair of launches
(lgcn_laneconv_fwd) for several unit groupings next to the one-launch lgcn_agg_mlp kernel.
Usage: python tools/bench_lc.py [--mma f16x2] [--scenes 32] [--groups 1,2,4] [--profile]   (--profile: plain loop for rocprofv3)"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lanegcn_amd  # noqa: F401,E402
from lanegcn_amd import data as gen  # noqa: E402
from lanegcn_amd import lanegcn as M  # noqa: E402
from lanegcn_amd import ops  # noqa: E402
from lanegcn_amd.engine import collate_flat  # noqa: E402
from tools.bench_agg import timeit  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mma", default="f16x2")
    ap.add_argument("--scenes", type=int, default=32)
    ap.add_argument("--groups", default="1:0,2:0,1:1,2:1,4:1", help="groups[:variant],...")
    ap.add_argument("--profile", action="store_true")
    args = ap.parse_args()
    ops.set_mma(args.mma)
    torch.manual_seed(0)
    net = M.MapNet(M.config).cuda().eval()
    fb = collate_flat(gen.synth_batch("S2", seed=100, n_scenes=args.scenes))
    with torch.no_grad():
        g64, _ = ops.graph_gather_indices(fb.idx_local, fb.seg_off, fb.seg_base)
        plan = ops.csr_build([g64[a:b] for (a, b), _ in fb.rel_slices], [g64[a:b] for _, (a, b) in fb.rel_slices], fb.n_nodes)
        x = torch.randn(fb.n_nodes, 128, device="cuda").relu()
        fuse, keys = net.fuse, M.rel_keys(6)
        wps = [ops.packed(fuse["ctr"][0].weight)] + [ops.packed(fuse[k][0].weight) for k in keys]
        c2 = fuse["ctr2"][0]
        gn1, wp2, gn2 = M._gn(fuse["norm"][0]), ops.packed(c2.linear.weight), M._gn(c2.norm)

        def fused():
            M.lane_conv({k: [v[0]] for k, v in fuse.items()}, x, plan, 6, impl="fused")

        res = {"fused": timeit(fused) if not args.profile else None}
        for g, var in [(int(v.split(":")[0]), int(v.split(":")[1]) if ":" in v else 0) for v in args.groups.split(",")]:
            lcp = ops.lc_plan(plan, n_groups=g if g > 0 else None, variant=var)
            part, out = ops.lc_part(lcp), torch.empty_like(x)
            fn = lambda: ops.laneconv_fwd(x, lcp, wps, gn1, wp2, gn2, part=part, out=out)
            if args.profile:
                for _ in range(20):
                    fn()
                torch.cuda.synchronize()
            else:
                res["tiled M=%d, %d group(s)" % (lcp.rows_per_block, len(lcp.gstart) - 1)] = timeit(fn)
        if args.profile:
            for _ in range(20):
                fused()
            torch.cuda.synchronize()
        for k, v in res.items():
            if v is not None:
                print("%-28s %8.2f us / layer" % (k, v))


if __name__ == "__main__":
    main()
