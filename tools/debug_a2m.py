import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import lanegcn_amd
from lanegcn_amd import lanegcn as M, ops
from conftest import to_torch_scene
from golden_io import load_scenes
from oracle import lanegcn_oracle as O
G = os.path.join(ROOT, "tests", "golden")
golden = dict(np.load(os.path.join(G, "hotpath_b4.npz")))
names_shapes = [(k, tuple(s)) for k, s in json.load(open(os.path.join(G, "state_dict_names.json")))]
sd = O.seeded_state(names_shapes, 7)
scenes = [to_torch_scene(s) for s in load_scenes(golden)]
g_cpu = O.graph_gather([s["graph"] for s in scenes])
feat0 = torch.from_numpy(golden["map_net"]); actors0 = torch.from_numpy(golden["actors_in"])
d_out = torch.randn(feat0.shape, generator=torch.Generator().manual_seed(0))
# oracle autograd (fp32 CPU)
sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k.startswith("a2m.")}
f_r, a_r = feat0.clone().requires_grad_(True), actors0.clone().requires_grad_(True)
out_r = O.a2m(f_r, g_cpu, a_r, [s["ctrs"] for s in scenes], sdr)
out_r.backward(d_out)
for mode in sys.argv[1:] or ["f32", "bf16x3"]:
    ops.set_mma(mode)
    mod = M.A2M(M.config); mod.load_state_dict({k[4:]: v for k, v in sd.items() if k.startswith("a2m.")}); mod = mod.cuda()
    graph = M.graph_gather([s["graph"] for s in scenes])
    f_d, a_d = feat0.cuda().requires_grad_(True), actors0.cuda().requires_grad_(True)
    actor_ctrs = [s["ctrs"].cuda() for s in scenes]
    idcs, st = [], 0
    for c in actor_ctrs:
        idcs.append(torch.arange(st, st + len(c), device="cuda")); st += len(c)
    out = mod(f_d, graph, a_d, idcs, actor_ctrs)
    out.backward(d_out.cuda())
    rel = lambda a, b: float((a.cpu() - b).abs().max() / (b.abs().max() + 1e-12))
    print("== mode", mode, "out", rel(out.detach(), out_r.detach()), "d feat", rel(f_d.grad, f_r.grad), "d actors", rel(a_d.grad, a_r.grad))
    worst = sorted(((rel(p.grad, sdr["a2m." + n].grad), n) for n, p in mod.named_parameters()), reverse=True)[:6]
    for e, n in worst: print("     %.2e %s" % (e, n))
