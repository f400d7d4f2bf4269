import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import lanegcn_amd
from lanegcn_amd import data as gen, ops
from lanegcn_amd.engine import HotPathEngine, collate_flat
import bench
dev = torch.device("cuda", 0)
mods = bench.build_modules(1234, dev)
eng = HotPathEngine(mods["map_net"], mods["a2m"], mods["m2m"], mods["m2a"], mods["a2a"])
lanes = []
for j in range(4):
    fb = collate_flat(gen.synth_batch("S2", seed=100 + 1000 * j), dev)
    a = torch.randn(fb.n_actors, 128, device=dev).relu()
    g, _ = eng.capture(fb, a)
    lanes.append((torch.cuda.Stream(), g))
torch.cuda.synchronize()
# host cost of one replay call (no sync)
for _ in range(20):
    for st, g in lanes:
        with torch.cuda.stream(st):
            g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter(); n = 0
host = []
for _ in range(50):
    for st, g in lanes:
        with torch.cuda.stream(st):
            a = time.perf_counter(); g.replay(); host.append(time.perf_counter() - a)
        n += 1
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print("replays %d: host issue %.3f ms each (median %.3f), wall %.3f ms each" % (n, t_issue / n * 1e3, np.median(host) * 1e3, t_all / n * 1e3))
