"""Throughput of LaneConv layers with several independent chains in flight (one captured graph of `layers`
layers per HIP stream, each on its own buffers): effective time per layer = wall / (graphs replayed * layers).
Usage: python tools/bench_lc_streams.py [--impl tiled --groups 1] [--streams 1,2,4] [--layers 8]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lanegcn_amd  # noqa: F401,E402
if "--stamps" in sys.argv:      # diagnostic library: in-kernel shader clock of k_lc_tile under load
    from lanegcn_amd import _lib as _L
    _L.LIB_PATH = os.path.join(os.path.dirname(_L.LIB_PATH), "liblgcn_stamps.so")
from lanegcn_amd import data as gen  # noqa: E402
from lanegcn_amd import lanegcn as M  # noqa: E402
from lanegcn_amd import ops  # noqa: E402
from lanegcn_amd.engine import collate_flat  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--impl", default="tiled")
    ap.add_argument("--groups", type=int, default=0)
    ap.add_argument("--streams", default="1,2,3,4")
    ap.add_argument("--layers", type=int, default=8)
    ap.add_argument("--mma", default="f16x2")
    ap.add_argument("--reps", type=int, default=60)
    ap.add_argument("--stamps", action="store_true")
    args = ap.parse_args()
    ops.set_mma(args.mma)
    ops.set_lc_groups(args.groups)
    torch.manual_seed(0)
    net = M.MapNet(M.config).cuda().eval()
    stamps = None
    if args.stamps:
        import ctypes
        from lanegcn_amd import _lib as L
        stamps = torch.zeros(4096 * 2 * 64, dtype=torch.int64, device="cuda")
        lib = L.load()
        lib.lgcn_debug_lc_stamps.argtypes = [ctypes.c_void_p]
        lib.lgcn_debug_lc_stamps(ctypes.c_void_p(stamps.data_ptr()))
    max_s = max(int(v) for v in args.streams.split(","))
    streams = [torch.cuda.Stream() for _ in range(max_s)]      # created first: consecutive pool streams = distinct hardware queues
    graphs = []
    for j in range(max_s):
        fb = collate_flat(gen.synth_batch("S2", seed=100 + j))
        with torch.no_grad():
            g64, _ = ops.graph_gather_indices(fb.idx_local, fb.seg_off, fb.seg_base)
            plan = ops.csr_build([g64[a:b] for (a, b), _ in fb.rel_slices], [g64[a:b] for _, (a, b) in fb.rel_slices], fb.n_nodes)
            x = torch.randn(fb.n_nodes, 128, device="cuda").relu()

            def body():
                y = x
                for _ in range(args.layers // 4):
                    y = M.lane_conv(net.fuse, y, plan, 6, impl=args.impl)
                return y

            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                body()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                out = body()
            graphs.append((g, out, plan, x))
    for n in [int(v) for v in args.streams.split(",")]:
        reps = args.reps
        for r in range(8):
            with torch.cuda.stream(streams[r % n]):
                graphs[r % n][0].replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for r in range(reps):
            with torch.cuda.stream(streams[r % n]):
                graphs[r % n][0].replay()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("%s groups=%d  %d stream(s): %.2f us per layer (effective)" % (args.impl, args.groups, n, dt / (reps * args.layers) * 1e6))
        if stamps is not None:
            import numpy as np
            st = stamps.cpu().numpy().reshape(-1, 64)
            st = st[st[:, 63] > 0]
            cyc = (st[:, 61] - st[:, 0]).astype(np.float64)
            ref = (st[:, 63] - st[:, 62]).astype(np.float64)
            ok = ref > 0
            print("    k_lc_tile workgroups: %d, median %.0f cycles in %.2f us -> shader clock %.2f GHz" % (
                ok.sum(), np.median(cyc[ok]), np.median(ref[ok]) / 100.0, np.median(cyc[ok] / ref[ok]) * 0.1))
            stamps.zero_()


if __name__ == "__main__":
    main()
