"""Generates tests/golden/*.npz by running the REFERENCE itself (leepaul009/LaneGCN-1 at
/root/reference, imported read-only with the in-process shims of SURVEY.md Appendix A) on small
synthetic scenes.  Run in the build container only:  python tests/golden/make_golden.py
The fixtures hold inputs + the reference's outputs (data), never reference source.
"""
import fractions
import json
import math
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.dont_write_bytecode = True          # never write into /root/reference


def import_reference():
    fractions.gcd = math.gcd                                   # lanegcn.py:8, layers.py:6 (py < 3.9)
    if not hasattr(np, "bool"):
        np.bool = bool                                         # data.py:167,206,521 (numpy < 1.24)

    def stub(name, **a):
        m = types.ModuleType(name)
        m.__dict__.update(a)
        sys.modules[name] = m

    for n in ("cv2", "argoverse", "argoverse.data_loading", "argoverse.map_representation", "skimage"):
        stub(n)
    stub("argoverse.data_loading.argoverse_forecasting_loader", ArgoverseForecastingLoader=object)
    stub("argoverse.map_representation.map_api", ArgoverseMap=object)
    stub("skimage.transform", rotate=None)
    sys.path.insert(0, "/root/reference")
    import lanegcn as ref                                      # noqa: E402
    import data as refdata                                     # noqa: E402
    ref.gpu = lambda x: x                                      # utils.gpu hard-calls .cuda()
    return ref, refdata


def main():
    import torch
    import lanegcn_amd  # noqa: F401  (our own generator; the reference only consumes its output)
    from lanegcn_amd import data as gen
    from golden_io import flatten
    from oracle.lanegcn_oracle import seeded_state

    torch.manual_seed(0)
    torch.set_num_threads(1)
    ref, refdata = import_reference()
    net = ref.Net(ref.config).eval()
    ref_sd = net.state_dict()
    shapes = [(k, tuple(v.shape)) for k, v in ref_sd.items()]
    with open(os.path.join(HERE, "state_dict_names.json"), "w") as f:
        json.dump([[k, list(s)] for k, s in shapes], f)

    SEED = 7
    net.load_state_dict(seeded_state(shapes, SEED))

    rng = np.random.default_rng(11)
    scenes = [gen.synth_scene(rng, [4, 6], 12), gen.synth_scene(rng, [4], 9),
              gen.synth_scene(rng, [5], 10), gen.synth_scene(rng, [4, 4], 11)]
    scenes[1]["ctrs"] = scenes[1]["ctrs"] + np.float32(1000.0)       # actors far from the map: zero A2M/M2A pairs
    for k in ("left", "right"):                                        # a scene without left/right edges
        scenes[2]["graph"][k] = {"u": np.zeros(0, np.int64), "v": np.zeros(0, np.int64)}

    out = {"seed": np.int64(SEED)}
    flatten(scenes, "scenes/", out)

    import copy
    batch = refdata.collate_fn(copy.deepcopy(scenes))
    with torch.no_grad():
        actors, actor_idcs = ref.actor_gather(batch["feats"])
        actor_ctrs = batch["ctrs"]
        actors = net.actor_net(actors)
        graph = ref.graph_gather(ref.to_long(batch["graph"]))
        out["actors_in"] = actors.numpy()
        for k1 in ("pre", "suc"):
            for i in range(6):
                for k2 in ("u", "v"):
                    out["gg/%s/%d/%s" % (k1, i, k2)] = graph[k1][i][k2].numpy()
        for k1 in ("left", "right"):
            for k2 in ("u", "v"):
                out["gg/%s/%s" % (k1, k2)] = graph[k1][k2].numpy()

        # pair sets exactly as Att.forward builds them (lanegcn.py:672-689)
        def pairs(agt_idcs, agt_ctrs, ctx_idcs, ctx_ctrs, th):
            hi, wi, hc, wc = [], [], 0, 0
            for i in range(len(agt_idcs)):
                dist = agt_ctrs[i].view(-1, 1, 2) - ctx_ctrs[i].view(1, -1, 2)
                dist = torch.sqrt((dist ** 2).sum(2))
                idcs = torch.nonzero(dist <= th, as_tuple=False)
                if len(idcs) == 0:
                    continue
                hi.append(idcs[:, 0] + hc)
                wi.append(idcs[:, 1] + wc)
                hc += len(agt_idcs[i])
                wc += len(ctx_idcs[i])
            return torch.cat(hi, 0).numpy(), torch.cat(wi, 0).numpy()

        cfg = ref.config
        out["pairs/a2m/hi"], out["pairs/a2m/wi"] = pairs(graph["idcs"], graph["ctrs"], actor_idcs, actor_ctrs, cfg["actor2map_dist"])
        out["pairs/m2a/hi"], out["pairs/m2a/wi"] = pairs(actor_idcs, actor_ctrs, graph["idcs"], graph["ctrs"], cfg["map2actor_dist"])
        out["pairs/a2a/hi"], out["pairs/a2a/wi"] = pairs(actor_idcs, actor_ctrs, actor_idcs, actor_ctrs, cfg["actor2actor_dist"])

        nodes, node_idcs, node_ctrs = net.map_net(graph)
        out["map_net"] = nodes.numpy().copy()
        nodes = net.a2m(nodes, graph, actors, actor_idcs, actor_ctrs)
        out["a2m"] = nodes.numpy().copy()
        nodes = net.m2m(nodes, graph)
        out["m2m"] = nodes.numpy().copy()
        act = net.m2a(actors, actor_idcs, actor_ctrs, nodes, node_idcs, node_ctrs)
        out["m2a"] = act.numpy().copy()
        act = net.a2a(act, actor_idcs, actor_ctrs)
        out["a2a"] = act.numpy().copy()

        # empty-context branch of Att (lanegcn.py:664-670)
        out["att_empty_ctx"] = net.a2m.att[0](out_t(out["map_net"]), graph["idcs"], graph["ctrs"],
                                              actors[:0], [], [], cfg["actor2map_dist"]).numpy().copy()

        # whole Net.forward (cls / reg), for the drop-in Net test
        full = net(refdata.collate_fn(copy.deepcopy(scenes)))
        for i in range(len(scenes)):
            out["net/cls/%d" % i] = full["cls"][i].numpy()
            out["net/reg/%d" % i] = full["reg"][i].numpy()

    np.savez_compressed(os.path.join(HERE, "hotpath_b4.npz"), **out)
    print("wrote hotpath_b4.npz:", {k: v.shape for k, v in out.items() if not k.startswith("scenes/")})


def out_t(a):
    import torch
    return torch.from_numpy(a.copy())


if __name__ == "__main__":
    main()
