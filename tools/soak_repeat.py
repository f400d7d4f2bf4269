"""Soak test of the bitwise repeatability of the captured forward with four lanes in flight: every
lane's graph is replayed N times round-robin on its own stream; after each replay a device-side comparison with the
lane's first result adds to a mismatch counter (no host synchronisation inside the loop).
Usage: python tools/soak_repeat.py [forwards per lane = 5000] [mma = f16x2] [lanes = 4] [library suffix, e.g. slp]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lanegcn_amd  # noqa: E402,F401
from lanegcn_amd import _lib as L  # noqa: E402

if len(sys.argv) > 4:      # a diagnostic build (make -C lanegcn-1_amd/csrc <suffix>) instead of the shipped library
    L.LIB_PATH = os.path.join(os.path.dirname(L.LIB_PATH), "liblgcn_%s.so" % sys.argv[4])
from lanegcn_amd import data as gen  # noqa: E402
from lanegcn_amd import lanegcn as M  # noqa: E402
from lanegcn_amd import ops  # noqa: E402
from lanegcn_amd.engine import HotPathEngine, collate_flat  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
    mma = sys.argv[2] if len(sys.argv) > 2 else "f16x2"
    S = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    ops.set_mma(mma)
    torch.manual_seed(0)
    mods = [cls(M.config).cuda().eval() for cls in (M.MapNet, M.A2M, M.M2M, M.M2A, M.A2A)]
    eng = HotPathEngine(*mods)
    streams = [torch.cuda.Stream() for _ in range(S)]
    lanes = []
    for j in range(S):
        fb = collate_flat(gen.synth_batch("S2", seed=300 + j))
        actors = torch.randn(fb.n_actors, 128, device="cuda").relu()
        g, out = eng.capture(fb, actors)
        g.replay()
        torch.cuda.synchronize()
        ref = (out["nodes"].clone(), out["actors"].clone())
        lanes.append((g, out, ref, torch.zeros(1, dtype=torch.int64, device="cuda")))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        for j, (g, out, ref, bad) in enumerate(lanes):
            with torch.cuda.stream(streams[j]):
                g.replay()
                bad += (out["nodes"] != ref[0]).any().to(torch.int64) + (out["actors"] != ref[1]).any().to(torch.int64)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    bad = [int(l[3].item()) for l in lanes]
    flags = [int(l[1]["nonfinite"].item()) for l in lanes]
    print("%s: mode %s, %d lanes x %d forwards (%.1f s, %.0f scenes/s incl. the comparisons): forwards whose nodes / actors "
          "differed from the lane's first result: %s; range-guard flags %s" % (os.path.basename(L.LIB_PATH), mma, S, n, dt, 32.0 * n * S / dt, bad, flags))
    return 1 if any(bad) else 0


if __name__ == "__main__":
    sys.exit(main())
