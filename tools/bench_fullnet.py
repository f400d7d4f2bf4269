"""End-to-end Net.forward (ActorNet + hot path + PredNet) on one S2 batch:
`eager` = drop-in Net.forward(data); `graph1` = FullNetEngine hipGraph replay, one forward at a time.
`graph4` = four whole-Net graphs in flight on four streams.  (Round 1 saw one GPU core dump in that mode and disabled
it; round 2 audited what the four captures share -- nothing mutable: each capture has its own memory pool, inputs and
MIOpen workspaces, the weight images are read-only -- and lgcn_gn_cl's bounds for L in {5, 10, 20} in both layouts,
found nothing, and ran it again once, clean: 0.63 ms per step, 51 k scenes/s end to end.  DESIGN.md section 5c.)"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lanegcn_amd  # noqa: E402,F401
from lanegcn_amd import data as gen  # noqa: E402
from lanegcn_amd import lanegcn as M  # noqa: E402
from lanegcn_amd.engine import FullNetEngine, collate_flat  # noqa: E402


def main():
    phase = sys.argv[1] if len(sys.argv) > 1 else "eager"      # eager | graph1 | graph4  (one phase per process)
    torch.manual_seed(0)
    net = M.Net(M.config).cuda().eval()
    res = {}
    scenes = gen.synth_batch("S2", seed=3)
    batch = gen.collate_fn(scenes)
    if phase == "eager":
        with torch.no_grad():
            for _ in range(3):
                net(batch)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                net(batch)
            torch.cuda.synchronize()
            res["eager_net_forward_ms"] = (time.perf_counter() - t0) / 10 * 1e3
        print(json.dumps(res))
        return
    n_lanes = 4 if phase == "graph4" else 1
    eng = FullNetEngine(net)
    lanes = []
    for j in range(n_lanes):
        sc = gen.synth_batch("S2", seed=3 + j)
        fb = collate_flat(sc)
        feats, rot, orig = eng.actor_inputs(sc)
        g, _ = eng.capture(fb, feats, rot, orig, [len(s["ctrs"]) for s in sc])
        lanes.append((torch.cuda.Stream(), g))      # (capture() keeps fb / feats / rot / orig alive with the graph)
    for n_streams in (n_lanes,):
        steps = 100
        for i in range(20):
            st, g = lanes[i % n_streams]
            with torch.cuda.stream(st):
                g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            st, g = lanes[i % n_streams]
            with torch.cuda.stream(st):
                g.replay()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        res["graph_%d_streams" % n_streams] = {"ms_per_step": ms, "scenes_per_s": 32e3 / ms}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
