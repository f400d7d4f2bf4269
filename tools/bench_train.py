"""Training-step timing (BASELINE.json config 3: end-to-end lanegcn.py training step, batch = 32, fp32):
Net forward + loss + backward + Adam step on one 32-scene synthetic batch (S2), HIP hot path fwd/bwd, against
the same step on the CPU with the oracle's stock-ATen statement of the hot path (reference-equivalent)."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lanegcn_amd  # noqa: E402,F401
from lanegcn_amd import data as gen  # noqa: E402
from lanegcn_amd import lanegcn as M  # noqa: E402
from lanegcn_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--mma", default=None)
    ap.add_argument("--cpu-steps", type=int, default=2)
    args = ap.parse_args()
    if args.mma:
        ops.set_mma(args.mma)
    torch.manual_seed(0)
    torch.autograd.set_multithreading_enabled(False)      # what train_dp.py does for its loop
    net = M.Net(M.config).cuda().train()
    loss_fn = M.Loss(M.config).cuda()
    opt = M.Optimizer(net.parameters(), M.config)
    batch = gen.collate_fn(gen.synth_batch("S2", seed=5))
    stages = {}

    def step(i, timed=False):
        marks = [time.perf_counter()]
        out = net(batch)
        if timed:
            torch.cuda.synchronize(); marks.append(time.perf_counter())
        lo = loss_fn(out, batch)
        opt.zero_grad()
        lo["loss"].backward()
        if timed:
            torch.cuda.synchronize(); marks.append(time.perf_counter())
        opt.step(i / 6436.0)
        if timed:
            torch.cuda.synchronize(); marks.append(time.perf_counter())
            for k, a, b in zip(("forward", "loss+backward", "adam"), marks[:-1], marks[1:]):
                stages.setdefault(k, []).append((b - a) * 1e3)
        return float(lo["loss"].detach())

    losses = [step(i) for i in range(args.warmup)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        losses.append(step(args.warmup + i))
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / args.steps * 1e3
    for i in range(5):
        step(args.warmup + args.steps + i, timed=True)
    res = {"metric": "training step (forward + loss + backward + Adam), batch 32, S2", "mma": ops.get_mma(),
           "ms_per_step": ms, "scenes_per_s": 32e3 / ms, "loss_first": losses[0], "loss_last": losses[-1],
           "stage_ms": {k: float(np.median(v)) for k, v in stages.items()}}
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
