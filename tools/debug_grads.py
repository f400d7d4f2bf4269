import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import lanegcn_amd
from lanegcn_amd import data as gen, lanegcn as M, ops
from golden_io import load_scenes
from oracle import lanegcn_oracle as O
ops.set_mma(sys.argv[1] if len(sys.argv) > 1 else "f32")
G = os.path.join(ROOT, "tests", "golden")
golden = dict(np.load(os.path.join(G, "hotpath_b4.npz")))
tg = dict(np.load(os.path.join(G, "train_b4.npz")))
names_shapes = [(k, tuple(s)) for k, s in json.load(open(os.path.join(G, "state_dict_names.json")))]
scenes = load_scenes(golden)
net = M.Net(M.config); net.load_state_dict(O.seeded_state(names_shapes, int(tg["seed"]))); net = net.cuda().train()
batch = gen.collate_fn(scenes)
out = net(batch); lo = M.Loss(M.config).cuda()(out, batch); lo["loss"].backward(); torch.cuda.synchronize()
names = json.load(open(os.path.join(G, "param_names.json")))
params = dict(net.named_parameters())
norms = np.array([float(params[n].grad.norm()) for n in names]); ref = tg["grad_norms"]
rel = np.abs(norms - ref) / (ref + 1e-12)
groups = {}
for n, r in zip(names, rel):
    groups.setdefault(n.split(".")[0], []).append(r)
for g, v in groups.items():
    print("%-10s params %3d  max rel norm err %.2e  median %.2e" % (g, len(v), max(v), float(np.median(v))))
order = np.argsort(-rel)[:12]
for i in order: print("   %-45s ours %.6g ref %.6g rel %.2e" % (names[i], norms[i], ref[i], rel[i]))
print("selected element-wise:")
rows = []
for k, v in tg.items():
    if k.startswith("grad/"):
        n = k[5:]; g = params[n].grad.cpu().numpy()
        rows.append((float(np.abs(g - v).max() / (np.abs(v).max() + 1e-12)), n))
for e, n in sorted(rows, reverse=True)[:40]: print("   %.2e %s" % (e, n))
