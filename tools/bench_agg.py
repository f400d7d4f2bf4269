"""Ablation micro-benchmark of the fused LaneConv launch (lgcn_agg_mlp) on the S2 graph.
Usage: python tools/bench_agg.py [mma ...]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lanegcn_amd  # noqa: F401,E402
from lanegcn_amd import _lib as L  # noqa: E402

# the work-skipping flag bits (1 << 8: no in-loop gathers, 1 << 9: no MFMA passes) and the LGCN_RB* / LGCN_RING
# environment knobs exist in the diagnostic library only (make -C lanegcn-1_amd/csrc ablate)
_ABLATE = os.path.join(os.path.dirname(L.LIB_PATH), "liblgcn_ablate.so")
if os.path.exists(_ABLATE):
    L.LIB_PATH = _ABLATE
from lanegcn_amd import data as gen  # noqa: E402
from lanegcn_amd import lanegcn as M  # noqa: E402
from lanegcn_amd import ops  # noqa: E402
from lanegcn_amd.engine import collate_flat  # noqa: E402


def timeit(fn, reps=20, rounds=5):
    """GPU time per call in us: `reps` calls captured in one hipGraph (no host launch overhead),
    best of `rounds` replays."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1e3)
    return best


def main():
    modes = sys.argv[1:] or ["f32", "bf16x3", "f16x2", "bf16"]
    torch.manual_seed(0)
    net = M.MapNet(M.config).cuda().eval()
    fb = collate_flat(gen.synth_batch("S2", seed=100))
    with torch.no_grad():
        g64, _ = ops.graph_gather_indices(fb.idx_local, fb.seg_off, fb.seg_base)
        us = [g64[a:b] for (a, b), _ in fb.rel_slices]
        vs = [g64[a:b] for _, (a, b) in fb.rel_slices]
        plan = ops.csr_build(us, vs, fb.n_nodes)
        N = fb.n_nodes
        x = torch.randn(N, 128, device="cuda")
        fuse = net.fuse
        keys = M.rel_keys(6)
        full = L.F_GN1 | L.F_RELU1 | L.F_GEMM2 | L.F_GN2 | L.F_RES | L.F_RELU2
        for mode in modes:
            ops.set_mma(mode)

            def rels(kind, n):
                out = [ops.RelSpec(x, ops.packed(fuse["ctr"][0].weight), L.REL_IDENT)]
                for r, key in enumerate(keys[: n - 1]):
                    w = ops.packed(fuse[key][0].weight)
                    out.append(ops.RelSpec(x, w, L.REL_CSR, r) if kind == "csr" else ops.RelSpec(x, w, L.REL_IDENT))
                return out

            c2 = fuse["ctr2"][0]
            kw = dict(gn1=(fuse["norm"][0].weight, fuse["norm"][0].bias), wp2=ops.packed(c2.linear.weight),
                      gn2=(c2.norm.weight, c2.norm.bias), res=x)
            out = torch.empty_like(x)

            def run(kind, n, flags=full, rb=0):
                r = rels(kind, n)
                extra = dict(rowptr=plan.rowptr, col=plan.col, n_rel_csr=plan.n_rel) if kind == "csr" else {}
                return timeit(lambda: ops.agg_mlp(N, r, flags, out=out, tile_rb=rb, **extra, **kw))

            print("== mma %s" % mode)
            rbs = (0,) if mode == "f32" else (1, 2, 3, 4)
            for rb in rbs:
                print("  laneconv 15 rel (CSR)   rb=%d : %7.1f us" % (rb, run("csr", 15, rb=rb)))
            for rb in rbs:
                print("  15 IDENT relations      rb=%d : %7.1f us" % (rb, run("ident", 15, rb=rb)))
            for n in (1, 2, 4, 8):
                print("  %2d IDENT relation(s)    rb=auto: %7.1f us" % (n, run("ident", n)))
            print("  1 IDENT, no GEMM2/GN    rb=auto: %7.1f us" % run("ident", 1, flags=0))
            if mode != "f32":
                for rb in (2, 3):
                    for kind in ("ident", "csr"):
                        print("  ABLATION rb=%d 15 %s: full %6.1f | no in-loop gather %6.1f | no MFMA %6.1f | neither %6.1f us" % (
                            rb, kind, run(kind, 15, rb=rb), run(kind, 15, flags=full | 256, rb=rb),
                            run(kind, 15, flags=full | 512, rb=rb), run(kind, 15, flags=full | 768, rb=rb)))
            print("  8 CSR relations         rb=auto: %7.1f us" % run("csr", 8))


if __name__ == "__main__":
    main()
