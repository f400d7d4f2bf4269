"""Round-2 investigation: run the S2 hot path in bf16x3 on the diagnostic library whose ReLU is the
compare + select form (`make -C lanegcn-1_amd/csrc relucnd`) -- the form under which three bf16x3 kernels once gave
wrong rows -- REPS times in ONE process, alone and with four forwards in flight, and compare every stage output with
the first run bit for bit (and the first run with the oracle).  Usage: python tools/relu_variant_check.py [cnd|max] [reps]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lanegcn_amd  # noqa: E402,F401
from lanegcn_amd import _lib as L  # noqa: E402

variant = sys.argv[1] if len(sys.argv) > 1 else "cnd"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
if variant == "cnd":
    L.LIB_PATH = os.path.join(os.path.dirname(L.LIB_PATH), "liblgcn_relucnd.so")
from lanegcn_amd import data as gen  # noqa: E402
from lanegcn_amd import lanegcn as M  # noqa: E402
from lanegcn_amd import ops  # noqa: E402
from lanegcn_amd.engine import HotPathEngine, collate_flat  # noqa: E402
from oracle import lanegcn_oracle as O  # noqa: E402  (checker)

ops.set_mma(sys.argv[3] if len(sys.argv) > 3 else "bf16x3")
ops.set_guard("off")
torch.manual_seed(0)
sd = O.seeded_state(O.hot_state_shapes(), 3)
mods = {}
for name, cls in (("map_net", M.MapNet), ("a2m", M.A2M), ("m2m", M.M2M), ("m2a", M.M2A), ("a2a", M.A2A)):
    m = cls(M.config)
    m.load_state_dict({k[len(name) + 1:]: v for k, v in sd.items() if k.startswith(name + ".")})
    mods[name] = m.cuda().eval()
scenes = gen.synth_batch("S2", seed=1)
fb = collate_flat(scenes)
actors_cpu = torch.from_numpy(np.random.default_rng(2).normal(0, 1, (fb.n_actors, 128)).astype(np.float32)).relu()
actors = actors_cpu.cuda()
eng = HotPathEngine(mods["map_net"], mods["a2m"], mods["m2m"], mods["m2a"], mods["a2a"], lane_impl="fused")
keys = ("map_net", "a2m", "m2m", "m2a", "a2a")
first = {k: v.cpu().numpy() for k, v in eng.forward(fb, actors, stages=True).items() if k in keys}
ts = [gen.from_numpy(s) for s in scenes]
want = O.hot_path(O.graph_gather([s["graph"] for s in ts]), actors_cpu, [s["ctrs"] for s in ts], sd)
print("library:", os.path.basename(L.LIB_PATH), "| first run vs oracle:",
      {k: float(np.abs(first[k] - want[k].numpy()).max()) for k in keys}, flush=True)


def compare(out, tag):
    bad = 0
    for k in keys:
        got = out[k].cpu().numpy()
        if not np.array_equal(got, first[k]):
            rows = np.nonzero((got != first[k]).any(1))[0]
            print("  %s: %s differs in %d rows (first %s, rows %% 64 -> %s), max |d| %.3g" %
                  (tag, k, len(rows), rows[:8], sorted(set((rows % 64).tolist()))[:16], float(np.abs(got - first[k]).max())), flush=True)
            bad += 1
    return bad


bad = 0
for r in range(reps):
    bad += compare(eng.forward(fb, actors, stages=True), "eager %d" % r)
print("eager, one stream: %d runs, %d stage outputs differ" % (reps, bad), flush=True)
streams = [torch.cuda.Stream() for _ in range(4)]
bad4 = 0
for r in range(reps // 4):
    outs = []
    for st in streams:
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            outs.append(eng.forward(fb, actors, stages=True) if st is streams[0] else
                        HotPathEngine(mods["map_net"], mods["a2m"], mods["m2m"], mods["m2a"], mods["a2a"], lane_impl="fused").forward(
                            collate_flat(scenes), actors, stages=True))
    torch.cuda.synchronize()
    for j, o in enumerate(outs):
        bad4 += compare(o, "4-stream round %d lane %d" % (r, j))
print("four forwards in flight: %d forwards, %d stage outputs differ" % (4 * (reps // 4), bad4), flush=True)
