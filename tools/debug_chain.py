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
actors0 = torch.from_numpy(golden["actors_in"])
gen = torch.Generator().manual_seed(0)
d_nodes, d_act = torch.randn(486, 128, generator=gen), torch.randn(42, 128, generator=gen)
hot = ("map_net.", "a2m.", "m2m.", "m2a.", "a2a.")
sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k.startswith(hot)}
a_r = actors0.clone().requires_grad_(True)
o_r = O.hot_path(g_cpu, a_r, [s["ctrs"] for s in scenes], sdr)
for k in ("map_net", "a2m", "m2m", "m2a"): o_r[k].retain_grad()
CUT = os.environ.get("CUT", "")
if CUT == "a2m":
    (o_r["a2m"] * d_nodes).sum().backward()
else:
    ((o_r["m2m"] * d_nodes).sum() + (o_r["a2a"] * d_act).sum()).backward()
rel = lambda a, b: float((a.cpu() - b).abs().max() / (b.abs().max() + 1e-12))
for mode in sys.argv[1:] or ["f32", "bf16x3"]:
    ops.set_mma(mode)
    mods = {}
    for name, cls in (("map_net", M.MapNet), ("a2m", M.A2M), ("m2m", M.M2M), ("m2a", M.M2A), ("a2a", M.A2A)):
        m = cls(M.config); m.load_state_dict({k[len(name) + 1:]: v for k, v in sd.items() if k.startswith(name + ".")}); mods[name] = m.cuda()
    graph = M.graph_gather([s["graph"] for s in scenes])
    a_d = actors0.cuda().requires_grad_(True)
    actor_ctrs = [s["ctrs"].cuda() for s in scenes]
    idcs, st = [], 0
    for c in actor_ctrs:
        idcs.append(torch.arange(st, st + len(c), device="cuda")); st += len(c)
    o = {}
    o["map_net"], nidcs, nctrs = mods["map_net"](graph)
    o["a2m"] = mods["a2m"](o["map_net"], graph, a_d, idcs, actor_ctrs)
    o["m2m"] = mods["m2m"](o["a2m"], graph)
    o["m2a"] = mods["m2a"](a_d, idcs, actor_ctrs, o["m2m"], nidcs, nctrs)
    o["a2a"] = mods["a2a"](o["m2a"], idcs, actor_ctrs)
    for k in ("map_net", "a2m", "m2m", "m2a"): o[k].retain_grad()
    if CUT == "a2m":
        (o["a2m"] * d_nodes.cuda()).sum().backward()
    else:
        ((o["m2m"] * d_nodes.cuda()).sum() + (o["a2a"] * d_act.cuda()).sum()).backward()
    print("== mode", mode)
    for k in (("a2m", "map_net") if CUT == "a2m" else ("a2a", "m2a", "m2m", "a2m", "map_net")):
        line = "  %-8s fwd %.1e" % (k, rel(o[k].detach(), o_r[k].detach()))
        if k != "a2a" and o[k].grad is not None and o_r[k].grad is not None: line += "  grad@out %.1e" % rel(o[k].grad, o_r[k].grad)
        worst = max((rel(p.grad, sdr[k + "." + n].grad), n) for n, p in mods[k].named_parameters())
        print(line, "  worst param grad %.1e (%s)" % worst)
    print("  d actors %.1e" % rel(a_d.grad, a_r.grad))
