import json, os, sys
import numpy as np, torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import lanegcn_amd
from lanegcn_amd import lanegcn as M, ops, autograd as A
from conftest import to_torch_scene
from golden_io import load_scenes
from oracle import lanegcn_oracle as O
G = os.path.join(ROOT, "tests", "golden")
golden = dict(np.load(os.path.join(G, "hotpath_b4.npz")))
names_shapes = [(k, tuple(s)) for k, s in json.load(open(os.path.join(G, "state_dict_names.json")))]
sd = O.seeded_state(names_shapes, 7)
scenes = [to_torch_scene(s) for s in load_scenes(golden)]
g_cpu = O.graph_gather([s["graph"] for s in scenes])
with torch.no_grad():
    ref_map = O.mapnet(g_cpu, sd)
    meta = torch.cat((g_cpu["turn"], g_cpu["control"].unsqueeze(1), g_cpu["intersect"].unsqueeze(1)), 1)
    pre_ref = F.linear(torch.cat((ref_map, meta), 1), sd["a2m.meta.linear.weight"])
    gn_ref = F.group_norm(pre_ref, 1, sd["a2m.meta.norm.weight"], sd["a2m.meta.norm.bias"], 1e-5)
for mode in ("f32", "bf16x3", "f16x2"):
    ops.set_mma(mode)
    mn = M.MapNet(M.config); mn.load_state_dict({k[8:]: v for k, v in sd.items() if k.startswith("map_net.")}); mn = mn.cuda()
    a2m = M.A2M(M.config); a2m.load_state_dict({k[4:]: v for k, v in sd.items() if k.startswith("a2m.")}); a2m = a2m.cuda()
    graph = M.graph_gather([s["graph"] for s in scenes])
    with torch.no_grad():
        feat, _, _ = mn(graph)
        out = a2m.fuse_meta(feat, graph["turn"], graph["control"], graph["intersect"]).cpu()
    flips = ((out > 0) != (gn_ref > 0))
    idx = flips.nonzero()
    print("mode %-7s meta-output ReLU mask mismatches vs oracle: %d of %d" % (mode, int(flips.sum()), flips.numel()),
          [(int(i), int(j), float(gn_ref[i, j])) for i, j in idx[:4]])
