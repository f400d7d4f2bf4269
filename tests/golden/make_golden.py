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
        restated = {"a2m": pairs(graph["idcs"], graph["ctrs"], actor_idcs, actor_ctrs, cfg["actor2map_dist"]),
                    "m2a": pairs(actor_idcs, actor_ctrs, graph["idcs"], graph["ctrs"], cfg["map2actor_dist"]),
                    "a2a": pairs(actor_idcs, actor_ctrs, actor_idcs, actor_ctrs, cfg["actor2actor_dist"])}

        # ... and the pair sets the reference's OWN Att.forward used: inside it the only torch.cat calls on int64
        # tensors are `hi = torch.cat(hi, 0)` and `wi = torch.cat(wi, 0)` (lanegcn.py:688-689), and hi is what
        # index_add_ receives (:703); both are recorded while the blocks run, nothing is re-derived
        captured = []
        real_cat, real_index_add = torch.cat, torch.Tensor.index_add_

        def spy_cat(tensors, *a, **k):
            r = real_cat(tensors, *a, **k)
            if r.dtype == torch.int64 and r.dim() == 1:
                captured.append(("cat", r.numpy().copy()))
            return r

        def spy_index_add(self, dim, index, source, *a, **k):
            captured.append(("index_add_", index.numpy().copy()))
            return real_index_add(self, dim, index, source, *a, **k)

        def run_block(fn, name):
            del captured[:]
            torch.cat, torch.Tensor.index_add_ = spy_cat, spy_index_add
            try:
                res = fn()
            finally:
                torch.cat, torch.Tensor.index_add_ = real_cat, real_index_add
            cats = [c for kind, c in captured if kind == "cat"]
            adds = [c for kind, c in captured if kind == "index_add_"]
            assert len(cats) == 4 and len(adds) == 2, (name, len(cats), len(adds))     # two Att layers per block
            hi, wi = cats[0], cats[1]
            assert np.array_equal(hi, cats[2]) and np.array_equal(wi, cats[3]) and np.array_equal(hi, adds[0])
            assert np.array_equal(hi, restated[name][0]) and np.array_equal(wi, restated[name][1]), name
            out["pairs/%s/hi" % name], out["pairs/%s/wi" % name] = hi, wi
            return res

        nodes, node_idcs, node_ctrs = net.map_net(graph)
        out["map_net"] = nodes.numpy().copy()
        nodes = run_block(lambda: net.a2m(nodes, graph, actors, actor_idcs, actor_ctrs), "a2m")
        out["a2m"] = nodes.numpy().copy()
        nodes = net.m2m(nodes, graph)
        out["m2m"] = nodes.numpy().copy()
        act = run_block(lambda: net.m2a(actors, actor_idcs, actor_ctrs, nodes, node_idcs, node_ctrs), "m2a")
        out["m2a"] = act.numpy().copy()
        act = run_block(lambda: net.a2a(act, actor_idcs, actor_ctrs), "a2a")
        out["a2a"] = act.numpy().copy()

        # data.dilated_nbrs of the reference itself (data.py:520-534, scipy csr products) on every scene's scale-0
        # pre / suc lists: scales 1..5 as sorted (u, v) edge sets (scipy fixes no order inside a row)
        for i, sc in enumerate(scenes):
            n = int(sc["graph"]["num_nodes"])
            for k1 in ("pre", "suc"):
                e0 = {k: np.asarray(v, np.int64) for k, v in sc["graph"][k1][0].items()}
                for j, d in enumerate(refdata.dilated_nbrs(e0, n, 6)):
                    uv = np.stack([np.asarray(d["u"], np.int64), np.asarray(d["v"], np.int64)], 1)
                    out["dil/%d/%s/%d" % (i, k1, j + 1)] = uv[np.lexsort((uv[:, 1], uv[:, 0]))]

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

    # ---- one training step of the reference: loss (lanegcn.py:740-821), backward, Adam step through the
    # reference's Optimizer (utils.py:98-162) at epoch 0 (lr 1e-3)
    tr = {"seed": np.int64(SEED)}
    net.zero_grad()
    batch = refdata.collate_fn(copy.deepcopy(scenes))
    loss_fn = ref.Loss(ref.config)
    outp = net(batch)
    loss_out = loss_fn(outp, batch)
    loss_out["loss"].backward()
    for k in ("cls_loss", "reg_loss", "loss"):
        tr["loss/" + k] = np.float64(loss_out[k].item())
    tr["loss/num_cls"] = np.int64(loss_out["num_cls"])
    tr["loss/num_reg"] = np.int64(loss_out["num_reg"])
    names = [k for k, _ in net.named_parameters()]
    tr["grad_norms"] = np.array([float(p.grad.norm()) if p.grad is not None else -1.0 for _, p in net.named_parameters()])
    picked = [n for n in names if n in SELECTED]
    assert len(picked) == len(SELECTED), set(SELECTED) - set(picked)
    params = dict(net.named_parameters())
    for n in picked:
        tr["grad/" + n] = params[n].grad.numpy().copy()
    opt = ref.Optimizer(net.parameters(), ref.config)
    lr = opt.step(0.0)
    tr["lr"] = np.float64(lr)
    for n in picked:     # post-step values of the small tensors + one matrix (the step itself is stock Adam)
        if params[n].numel() <= 512 or n == "map_net.fuse.ctr.0.weight":
            tr["after/" + n] = params[n].detach().numpy().copy()
    with open(os.path.join(HERE, "param_names.json"), "w") as f:
        json.dump(names, f)
    np.savez_compressed(os.path.join(HERE, "train_b4.npz"), **tr)
    print("wrote train_b4.npz: loss %.6f cls %.6f reg %.6f num_cls %d num_reg %d" % (
        tr["loss/loss"], tr["loss/cls_loss"], tr["loss/reg_loss"], tr["loss/num_cls"], tr["loss/num_reg"]))
    train_b32(ref, refdata, shapes, SEED)
    lanercnn_fixture(SEED)


def train_b32(ref, refdata, shapes, seed):
    """BASELINE config 4: one training step of the reference at batch 32 (workload S2).  The batch is regenerated
    from its seed on the other side; the fixture holds the loss, every parameter's gradient norm and, for the
    hot-path tensors of SELECTED, the gradient itself (matrices: every 8th row)."""
    import copy
    import torch
    from lanegcn_amd import data as gen
    from oracle.lanegcn_oracle import seeded_state
    torch.set_num_threads(8)
    net = ref.Net(ref.config).train()
    net.load_state_dict(seeded_state(shapes, seed))
    scenes = gen.synth_batch("S2", seed=B32_BATCH_SEED)
    batch = refdata.collate_fn(copy.deepcopy(scenes))
    loss_out = ref.Loss(ref.config)(net(batch), batch)
    loss_out["loss"].backward()
    tr = {"seed": np.int64(seed), "batch_seed": np.int64(B32_BATCH_SEED)}
    for k in ("cls_loss", "reg_loss", "loss"):
        tr["loss/" + k] = np.float64(loss_out[k].item())
    tr["loss/num_cls"] = np.int64(loss_out["num_cls"])
    tr["loss/num_reg"] = np.int64(loss_out["num_reg"])
    tr["grad_norms"] = np.array([float(p.grad.norm()) if p.grad is not None else -1.0 for _, p in net.named_parameters()])
    params = dict(net.named_parameters())
    for n in SELECTED:
        g = params[n].grad.numpy()
        tr["grad/" + n] = (g[::8] if g.ndim == 2 and g.shape[0] == 128 and g.shape[1] >= 128 else g).copy()
    np.savez_compressed(os.path.join(HERE, "train_b32.npz"), **tr)
    print("wrote train_b32.npz: loss %.6f cls %.6f reg %.6f num_cls %d num_reg %d" % (
        tr["loss/loss"], tr["loss/cls_loss"], tr["loss/reg_loss"], tr["loss/num_cls"], tr["loss/num_reg"]))


def lanercnn_fixture(seed):
    """Row f4: the graph modules of the reference's fork model (lanercnn.py: LaneInput 280-351, LaneRoI 354-430,
    LanePooling 433-514, GlobalGraphNet 517-600) run by the reference itself on small synthetic lane graphs."""
    import types
    import torch
    from lanegcn_amd import data as gen
    from golden_io import flatten
    if "torchvision" not in sys.modules:
        try:
            import torchvision  # noqa: F401
        except Exception:                                   # not installed here; lanercnn.py only imports it
            sys.modules["torchvision"] = types.ModuleType("torchvision")
    import lanercnn as rl
    torch.set_num_threads(1)
    rng = np.random.default_rng(29)
    scenes = [gen.synth_scene(rng, [4, 3], 6), gen.synth_scene(rng, [5], 4), gen.synth_scene(rng, [3, 3], 5)]
    graphs = [s["graph"] for s in scenes]
    import lanegcn as ref
    import data as refdata
    import copy
    g = ref.graph_gather(ref.to_long(refdata.collate_fn(copy.deepcopy(scenes))["graph"]))
    n = g["feats"].shape[0]
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))

    from oracle.lanercnn_oracle import seeded_state as seeded_rcnn
    out = {"seed": np.int64(seed)}
    flatten(scenes, "scenes/", out)
    mods = {"roi": rl.LaneRoI(rl.config, 128), "ggn": rl.GlobalGraphNet(rl.config), "pool": rl.LanePooling(128, 128),
            "input": rl.LaneInput(rl.config)}
    names = {}
    for i, (name, m) in enumerate(mods.items()):
        shapes = [(k, tuple(v.shape)) for k, v in m.state_dict().items()]
        m.eval().load_state_dict(seeded_rcnn(shapes, seed + i))      # weights are regenerated on the other side
        names[name] = [[k, list(sh)] for k, sh in shapes]
    with open(os.path.join(HERE, "lanercnn_state_names.json"), "w") as f:
        json.dump(names, f)
    with torch.no_grad():
        x = t(rng.normal(0, 1, (n, 128)).astype(np.float32))
        out["roi/x"] = x.numpy()
        out["roi/out"] = mods["roi"](x.clone(), g).numpy().copy()
        out["ggn/out"] = mods["ggn"](F_relu(x.clone()), g).numpy().copy()
        # LaneInput: 8-d node features, 80-d agent features, agent -> node edges
        n_agt = [5, 3, 4]
        counts = np.cumsum([0] + [int(gr["num_nodes"]) for gr in graphs])
        acount = np.cumsum([0] + n_agt)
        li = {"feats": [t(rng.normal(0, 1, (int(gr["num_nodes"]), 8)).astype(np.float32)) for gr in graphs],
              "agent_feat": [t(rng.normal(0, 1, (a, 80)).astype(np.float32)) for a in n_agt]}
        u, v = [], []
        for i, gr in enumerate(graphs):
            m_e = 3 * int(gr["num_nodes"])
            u.append(rng.integers(0, n_agt[i], m_e) + acount[i])
            v.append(rng.integers(0, int(gr["num_nodes"]), m_e) + counts[i])
        li["a2m"] = {"u": t(np.concatenate(u).astype(np.int64)), "v": t(np.concatenate(v).astype(np.int64))}
        for i in range(3):
            out["li/feats/%d" % i], out["li/agent_feat/%d" % i] = li["feats"][i].numpy(), li["agent_feat"][i].numpy()
        out["li/a2m/u"], out["li/a2m/v"] = li["a2m"]["u"].numpy(), li["a2m"]["v"].numpy()
        out["li/out"] = mods["input"](li).numpy().copy()
        # LanePooling: context = every scene's lane graph, target = a subset of its nodes (scene 1 far away: no pairs)
        ctx_g = {"ctrs": [t(gr["ctrs"].astype(np.float32)) for gr in graphs],
                 "pose": [t(np.concatenate([gr["ctrs"], gr["feats"]], 1).astype(np.float32)) for gr in graphs]}
        tgt_ctrs, tgt_pose = [], []
        for i, gr in enumerate(graphs):
            pick = rng.choice(int(gr["num_nodes"]), 20, replace=False)
            c = gr["ctrs"][pick].astype(np.float32) + rng.normal(0, 1.0, (20, 2)).astype(np.float32)
            if i == 1:
                c = c + np.float32(500.0)
            tgt_ctrs.append(t(c))
            tgt_pose.append(t(np.concatenate([c, gr["feats"][pick].astype(np.float32)], 1)))
        tgt_g = {"ctrs": tgt_ctrs, "pose": tgt_pose}
        cfeat = F_relu(t(rng.normal(0, 1, (n, 128)).astype(np.float32)))
        tfeat = F_relu(t(rng.normal(0, 1, (60, 128)).astype(np.float32)))
        for i in range(3):
            out["pool/tgt_ctrs/%d" % i], out["pool/tgt_pose/%d" % i] = tgt_ctrs[i].numpy(), tgt_pose[i].numpy()
        out["pool/cfeat"], out["pool/tfeat"] = cfeat.numpy(), tfeat.numpy()
        captured = []
        real_index_add = torch.Tensor.index_add_

        def spy(self, dim, index, source, *a, **k):
            captured.append(index.numpy().copy())
            return real_index_add(self, dim, index, source, *a, **k)

        torch.Tensor.index_add_ = spy
        try:
            out["pool/out"] = mods["pool"](cfeat, ctx_g, tfeat.clone(), tgt_g, 6.0).numpy().copy()
        finally:
            torch.Tensor.index_add_ = real_index_add
        assert len(captured) == 1
        out["pool/wi"] = captured[0]
    np.savez_compressed(os.path.join(HERE, "lanercnn_b3.npz"), **out)
    print("wrote lanercnn_b3.npz: %d nodes, pooling pairs %d" % (n, len(out["pool/wi"])))


def F_relu(x):
    import torch
    return torch.relu(x)


B32_BATCH_SEED = 41

SELECTED = [
    "actor_net.groups.0.0.conv1.weight", "actor_net.output.conv2.weight",
    "map_net.input.0.weight", "map_net.input.0.bias", "map_net.input.2.linear.weight", "map_net.input.2.norm.weight",
    "map_net.seg.2.norm.bias", "map_net.fuse.ctr.0.weight", "map_net.fuse.pre3.2.weight", "map_net.fuse.left.1.weight",
    "map_net.fuse.suc0.3.weight", "map_net.fuse.norm.3.weight", "map_net.fuse.norm.3.bias",
    "map_net.fuse.ctr2.0.linear.weight", "map_net.fuse.ctr2.2.norm.weight",
    "a2m.meta.linear.weight", "a2m.meta.norm.bias", "a2m.att.0.dist.0.weight", "a2m.att.0.dist.0.bias",
    "a2m.att.0.dist.2.linear.weight", "a2m.att.0.dist.2.norm.weight", "a2m.att.0.query.linear.weight",
    "a2m.att.0.ctx.0.linear.weight", "a2m.att.0.ctx.0.norm.bias", "a2m.att.0.ctx.1.weight", "a2m.att.0.agt.weight",
    "a2m.att.0.norm.weight", "a2m.att.0.linear.linear.weight", "a2m.att.1.linear.norm.bias",
    "m2m.fuse.suc5.3.weight", "m2m.fuse.ctr.1.weight", "m2m.fuse.right.0.weight",
    "m2a.att.1.ctx.0.linear.weight", "m2a.att.0.query.norm.weight", "a2a.att.0.agt.weight", "a2a.att.1.query.norm.weight",
    "a2a.att.1.ctx.1.weight", "pred_net.cls.1.weight", "pred_net.att_dest.dist.2.linear.weight",
]


def out_t(a):
    import torch
    return torch.from_numpy(a.copy())


def graphgen_scene(rng, n_lanes, seg_per_lane, side_p=1.0, far=False):
    """Raw lane topology of one synthetic scene (what ArgoDataset.get_lane_graph hands to `preprocess`, reference
    data.py:220-361): `n_lanes` parallel lanes of a road bent along an arc, each cut into consecutive LANE SEGMENTS
    (the reference's "lanes") of `seg_per_lane` nodes; segment k of lane j follows segment k - 1 (pre / suc pairs) and
    has lane j - 1 / j + 1's segment k as left / right neighbour (kept with probability side_p)."""
    n_seg = int(rng.integers(2, 5))
    ctrs, feats, lane_idcs = [], [], []
    pre_pairs, suc_pairs, left_pairs, right_pairs = [], [], [], []
    radius = float(rng.uniform(40.0, 200.0))
    lane_id = lambda j, k: j * n_seg + k
    for j in range(n_lanes):
        r = radius + 3.5 * j * (40.0 if far else 1.0)
        m = n_seg * seg_per_lane + 1
        ang = np.linspace(0.0, 2.0 * m / radius, m + 1) + rng.normal(0, 1e-3, m + 1)
        pts = np.stack([r * np.cos(ang), r * np.sin(ang)], 1)
        c, f = (pts[:-1] + pts[1:]) / 2, pts[1:] - pts[:-1]
        for k in range(n_seg):
            sl = slice(k * seg_per_lane, (k + 1) * seg_per_lane)
            ctrs.append(c[sl]); feats.append(f[sl])
            lane_idcs.append(np.full(seg_per_lane, lane_id(j, k)))
            if k > 0:
                pre_pairs.append([lane_id(j, k), lane_id(j, k - 1)])
                suc_pairs.append([lane_id(j, k - 1), lane_id(j, k)])
            if j > 0 and rng.random() < side_p:
                right_pairs.append([lane_id(j, k), lane_id(j - 1, k)])
            if j + 1 < n_lanes and rng.random() < side_p:
                left_pairs.append([lane_id(j, k), lane_id(j + 1, k)])
    arr = lambda x: np.asarray(x, np.int64).reshape(-1, 2)
    return dict(ctrs=np.concatenate(ctrs).astype(np.float32), feats=np.concatenate(feats).astype(np.float32),
                lane_idcs=np.concatenate(lane_idcs).astype(np.int64), pre_pairs=arr(pre_pairs), suc_pairs=arr(suc_pairs),
                left_pairs=arr(left_pairs), right_pairs=arr(right_pairs))


def graphgen_fixture():
    """Row f3: the reference's own `preprocess` (preprocess_data.py:287-392, cross_angle = None, cross_dist = 6 as
    lanegcn.py's config) on six synthetic raw lane topologies; inputs and the left / right edges it returns."""
    import torch
    import_reference()
    import preprocess_data as refpp
    from oracle import graphgen_oracle as GO
    torch.set_num_threads(1)
    rng = np.random.default_rng(41)
    specs = [dict(n_lanes=3, seg_per_lane=9), dict(n_lanes=5, seg_per_lane=6, side_p=0.6), dict(n_lanes=2, seg_per_lane=12),
             dict(n_lanes=4, seg_per_lane=7, far=True), dict(n_lanes=6, seg_per_lane=5, side_p=0.8), dict(n_lanes=1, seg_per_lane=8)]
    out = {"cross_dist": np.float32(6.0), "n_scenes": np.int64(len(specs))}
    for i, sp in enumerate(specs):
        g = graphgen_scene(rng, **sp)
        if i == 2:
            g["left_pairs"] = np.zeros((0, 2), np.int64)               # the empty-side branch (:348-350)
        tg = {k: torch.from_numpy(v) for k, v in g.items()}
        tg["idx"] = i
        res = refpp.preprocess(tg, 6.0)
        mine = GO.preprocess(g, 6.0)
        for side in ("left", "right"):
            u, v = res[side]["u"], res[side]["v"]
            assert u.dtype == np.int16 and np.array_equal(u, mine[side]["u"]) and np.array_equal(v, mine[side]["v"]), (i, side)
            out["g%d/%s/u" % (i, side)], out["g%d/%s/v" % (i, side)] = u, v
        for k, v in g.items():
            out["g%d/%s" % (i, k)] = v
        print("scene %d: %d nodes, %d lanes, left %d right %d edges" % (
            i, len(g["lane_idcs"]), int(g["lane_idcs"][-1]) + 1, len(res["left"]["u"]), len(res["right"]["u"])))
    np.savez_compressed(os.path.join(HERE, "graphgen_b6.npz"), **out)


BIG_SEED = 77


def big_inputs(seed=BIG_SEED):
    """Inputs of the at-size fixtures, regenerated identically by the tests (tests/test_big_fixtures.py imports this
    function): two lane-graph scenes of 1,080 nodes each, node / target features, two raw lane topologies of >= 1 k nodes."""
    from lanegcn_amd import data as gen
    rng = np.random.default_rng(seed)
    scenes = [gen.synth_scene(rng, [6] * 10, 12), gen.synth_scene(rng, [6] * 10, 9)]
    n = sum(int(s["graph"]["num_nodes"]) for s in scenes)
    x = rng.normal(0, 1, (n, 128)).astype(np.float32)
    cfeat = np.maximum(rng.normal(0, 1, (n, 128)), 0).astype(np.float32)
    tgt = []
    for sc in scenes:
        gr = sc["graph"]
        pick = rng.choice(int(gr["num_nodes"]), 200, replace=False)
        c = gr["ctrs"][pick].astype(np.float32) + rng.normal(0, 1.0, (200, 2)).astype(np.float32)
        tgt.append((c, np.concatenate([c, gr["feats"][pick].astype(np.float32)], 1)))
    tfeat = np.maximum(rng.normal(0, 1, (400, 128)), 0).astype(np.float32)
    raw = [graphgen_scene(rng, n_lanes=12, seg_per_lane=40), graphgen_scene(rng, n_lanes=9, seg_per_lane=48, side_p=0.7)]
    return scenes, x, cfeat, tgt, tfeat, raw


def edge_hash(u, v):
    import hashlib
    uv = np.stack([np.asarray(u, np.int64), np.asarray(v, np.int64)], 1)
    uv = uv[np.lexsort((uv[:, 1], uv[:, 0]))]
    return np.frombuffer(hashlib.sha1(np.ascontiguousarray(uv).tobytes()).digest(), np.uint8).copy(), len(uv)


def big_fixture(seed=3):
    """Rows f3 / f4 AT SIZE (>= 1 k nodes per scene): the reference's own preprocess, data.dilated_nbrs and lanercnn
    modules on the inputs of big_inputs().  Stored: integer outputs in full (int16 left / right arrays) or as SHA-1 of the
    sorted edge set (dilated relations, pooling index); feature outputs as every 16th row plus float64 column sums."""
    import copy
    import types
    import torch
    import_reference()
    if "torchvision" not in sys.modules:
        sys.modules["torchvision"] = types.ModuleType("torchvision")
    import data as refdata
    import lanegcn as ref
    import lanercnn as rl
    import preprocess_data as refpp
    from oracle.lanercnn_oracle import seeded_state as seeded_rcnn
    torch.set_num_threads(4)
    scenes, x, cfeat, tgt, tfeat, raw = big_inputs()
    out = {"seed": np.int64(seed)}
    # f3: left / right edges of the raw topologies; dilated pre / suc of the first scene's scale-0 relations
    for i, g in enumerate(raw):
        tg = {k: torch.from_numpy(v) for k, v in g.items()}
        tg["idx"] = i
        res = refpp.preprocess(tg, 6.0)
        for side in ("left", "right"):
            out["pp%d/%s/u" % (i, side)], out["pp%d/%s/v" % (i, side)] = res[side]["u"], res[side]["v"]
        print("raw topology %d: %d nodes, left %d right %d" % (i, len(g["lane_idcs"]), len(res["left"]["u"]), len(res["right"]["u"])))
    gr = scenes[0]["graph"]
    for k1 in ("pre", "suc"):
        nb = refdata.dilated_nbrs({"u": gr[k1][0]["u"], "v": gr[k1][0]["v"]}, int(gr["num_nodes"]), 6)
        for j, e in enumerate(nb):
            h, cnt = edge_hash(e["u"], e["v"])
            out["dil/%s/%d/sha1" % (k1, j + 1)], out["dil/%s/%d/n" % (k1, j + 1)] = h, np.int64(cnt)
    # f4: the fork model's graph modules
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    g = ref.graph_gather(ref.to_long(refdata.collate_fn(copy.deepcopy(scenes))["graph"]))
    mods = {"roi": rl.LaneRoI(rl.config, 128), "ggn": rl.GlobalGraphNet(rl.config), "pool": rl.LanePooling(128, 128)}
    for i, (name, m) in enumerate(mods.items()):
        m.eval().load_state_dict(seeded_rcnn([(k, tuple(v.shape)) for k, v in m.state_dict().items()], seed + i))

    def keep(key, a):
        a = a.numpy()
        out[key + "/rows16"], out[key + "/colsum"] = a[::16].copy(), a.astype(np.float64).sum(0)

    with torch.no_grad():
        keep("roi", mods["roi"](t(x).clone(), g))
        keep("ggn", mods["ggn"](torch.relu(t(x).clone()), g))
        graphs = [s["graph"] for s in scenes]
        ctx_g = {"ctrs": [t(gr["ctrs"].astype(np.float32)) for gr in graphs],
                 "pose": [t(np.concatenate([gr["ctrs"], gr["feats"]], 1).astype(np.float32)) for gr in graphs]}
        tgt_g = {"ctrs": [t(c) for c, _ in tgt], "pose": [t(p_) for _, p_ in tgt]}
        captured = []
        real = torch.Tensor.index_add_

        def spy(self, dim, index, source, *a, **k):
            captured.append(index.numpy().copy())
            return real(self, dim, index, source, *a, **k)

        torch.Tensor.index_add_ = spy
        try:
            keep("pool", mods["pool"](t(cfeat), ctx_g, t(tfeat).clone(), tgt_g, 6.0))
        finally:
            torch.Tensor.index_add_ = real
        import hashlib
        out["pool/wi_sha1"] = np.frombuffer(hashlib.sha1(np.ascontiguousarray(captured[0].astype(np.int64)).tobytes()).digest(), np.uint8).copy()
        out["pool/n_pairs"] = np.int64(len(captured[0]))
    np.savez_compressed(os.path.join(HERE, "big_f3f4.npz"), **out)
    print("wrote big_f3f4.npz: %d nodes, pooling pairs %d" % (x.shape[0], int(out["pool/n_pairs"])))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "big":
        big_fixture()
    elif len(sys.argv) > 1 and sys.argv[1] == "graphgen":
        graphgen_fixture()
    else:
        main()
        graphgen_fixture()
