"""How well do several captured forwards overlap on the GPU, and does the host block in graph launches?
Usage: python tools/bench_overlap.py [--impl fused|tiled] [--groups N]"""
import argparse
import os
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lanegcn_amd  # noqa: F401,E402
from lanegcn_amd import data as gen  # noqa: E402
from lanegcn_amd import ops  # noqa: E402
from lanegcn_amd.engine import HotPathEngine, collate_flat  # noqa: E402
from bench import build_modules  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--impl", default="fused")
    ap.add_argument("--groups", type=int, default=0)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--only4", action="store_true", help="only the 4 graphs on 4 streams case (for rocprofv3)")
    ap.add_argument("--streams-first", action="store_true", help="create the replay streams before capturing")
    ap.add_argument("--n", type=int, default=8, help="graphs / streams to create")
    args = ap.parse_args()
    ops.set_laneconv_impl(args.impl)
    ops.set_lc_groups(args.groups)
    dev = torch.device("cuda", 0)
    mods = build_modules(1234, dev)
    eng = HotPathEngine(mods["map_net"], mods["a2m"], mods["m2m"], mods["m2a"], mods["a2a"])

    def lane(j):
        fb = collate_flat(gen.synth_batch("S2", seed=100 + 1000 * j), dev)
        a = torch.randn(fb.n_actors, 128, device=dev).relu()
        g, _ = eng.capture(fb, a)
        return g

    if args.streams_first:
        streams = [torch.cuda.Stream() for _ in range(args.n)]
        graphs = [lane(j) for j in range(args.n)]
    else:
        graphs = [lane(j) for j in range(args.n)]
        streams = [torch.cuda.Stream() for _ in range(args.n)]
    print("stream handles:", [hex(s_.cuda_stream) for s_ in streams])

    def run(assign, steps, label):
        # assign: list of (graph index, stream index) visited round-robin
        for k in range(16):
            gi, si = assign[k % len(assign)]
            with torch.cuda.stream(streams[si]):
                graphs[gi].replay()
        torch.cuda.synchronize()
        host = []
        t0 = time.perf_counter()
        for k in range(steps):
            gi, si = assign[k % len(assign)]
            h0 = time.perf_counter()
            with torch.cuda.stream(streams[si]):
                graphs[gi].replay()
            host.append(time.perf_counter() - h0)
        t_issue = time.perf_counter() - t0
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        host = np.array(host) * 1e6
        print("%-44s %7.0f scenes/s  %.3f ms/step | host per replay: median %.0f us, p95 %.0f, max %.0f; issue loop %.1f ms of %.1f" % (
            label, 32 * steps / dt, dt / steps * 1e3, np.median(host), np.percentile(host, 95), host.max(), t_issue * 1e3, dt * 1e3))

    S = args.steps
    if args.only4:
        run([(j, j) for j in range(4)], S, "4 graphs on 4 streams")
        return
    run([(0, 0)], S, "1 graph, 1 stream")
    run([(0, 0), (1, 0)], S, "2 graphs, 1 stream")
    run([(j, j) for j in range(2)], S, "2 graphs on 2 streams")
    run([(j, j) for j in range(4)], S, "4 graphs on 4 streams")
    run([(j, j % 4) for j in range(8)], S, "8 graphs on 4 streams")
    run([(j, j) for j in range(8)], S, "8 graphs on 8 streams")

    def threaded(n, steps):
        for j in range(n):
            with torch.cuda.stream(streams[j]):
                graphs[j].replay()
        torch.cuda.synchronize()
        per = steps // n

        def work(j):
            with torch.cuda.stream(streams[j]):
                for _ in range(per):
                    graphs[j].replay()

        ths = [threading.Thread(target=work, args=(j,)) for j in range(n)]
        t0 = time.perf_counter()
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("%-44s %7.0f scenes/s  %.3f ms/step" % ("%d host threads, one graph + stream each" % n, 32 * per * n / dt, dt / (per * n) * 1e3))

    threaded(4, S)
    threaded(8, S)


if __name__ == "__main__":
    main()
