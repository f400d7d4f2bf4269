"""Row f4: the graph modules of the reference's fork model lanercnn.py.

CPU: the oracle restatement against the reference's own outputs (tests/golden/lanercnn_b3.npz, produced by
make_golden.py from the imported reference) and the product's state_dict layout against the reference's.
GPU: the product's modules (HIP kernels) against the same captures, in every matrix mode that claims fp32 parity."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR, to_torch_scene
from golden_io import load_scenes
from oracle import lanegcn_oracle as O
from oracle import lanercnn_oracle as OR

FTOL = 1e-4


@pytest.fixture(scope="module")
def fx():
    with np.load(os.path.join(GOLDEN_DIR, "lanercnn_b3.npz")) as z:
        g = {k: z[k] for k in z.files}
    names = json.load(open(os.path.join(GOLDEN_DIR, "lanercnn_state_names.json")))
    return g, names


def inputs(g, device=None):
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device) if device else torch.from_numpy(np.ascontiguousarray(a))
    scenes = [to_torch_scene(s) for s in load_scenes(g)]
    li = {"feats": [dev(g["li/feats/%d" % i]) for i in range(3)], "agent_feat": [dev(g["li/agent_feat/%d" % i]) for i in range(3)],
          "a2m": {"u": dev(g["li/a2m/u"]), "v": dev(g["li/a2m/v"])}}
    ctx_g = {"ctrs": [dev(s["graph"]["ctrs"].numpy().astype(np.float32)) for s in scenes],
             "pose": [dev(np.concatenate([s["graph"]["ctrs"].numpy(), s["graph"]["feats"].numpy()], 1).astype(np.float32)) for s in scenes]}
    tgt_g = {"ctrs": [dev(g["pool/tgt_ctrs/%d" % i]) for i in range(3)], "pose": [dev(g["pool/tgt_pose/%d" % i]) for i in range(3)]}
    return scenes, li, ctx_g, tgt_g, dev


def sd_for(names, key, seed, idx, prefix):
    sd = OR.seeded_state([(k, tuple(s)) for k, s in names[key]], seed + idx)
    return {prefix + "." + k: v for k, v in sd.items()}, sd


def test_oracle_vs_reference_captures(fx):
    g, names = fx
    seed = int(g["seed"])
    scenes, li, ctx_g, tgt_g, dev = inputs(g)
    graph = O.graph_gather([s["graph"] for s in scenes])
    x = torch.from_numpy(g["roi/x"])
    sd, _ = sd_for(names, "roi", seed, 0, "roi")
    assert float((OR.lane_roi(x, graph, sd) - torch.from_numpy(g["roi/out"])).abs().max()) <= 1e-5
    sd, _ = sd_for(names, "ggn", seed, 1, "ggn")
    assert float((OR.global_graph_net(torch.relu(x), graph, sd) - torch.from_numpy(g["ggn/out"])).abs().max()) <= 1e-5
    sd, _ = sd_for(names, "pool", seed, 2, "pool")
    out, hi, wi = OR.lane_pooling(torch.from_numpy(g["pool/cfeat"]), ctx_g, torch.from_numpy(g["pool/tfeat"]), tgt_g, sd)
    assert np.array_equal(wi.numpy(), g["pool/wi"])                      # the index the reference's index_add_ received
    assert float((out - torch.from_numpy(g["pool/out"])).abs().max()) <= 1e-5
    sd, _ = sd_for(names, "input", seed, 3, "input")
    assert float((OR.lane_input(li, sd) - torch.from_numpy(g["li/out"])).abs().max()) <= 1e-5


def test_state_dict_layout_is_the_reference_layout(fx):
    _, names = fx
    import lanegcn_amd  # noqa: F401
    from lanegcn_amd import lanegcn as M
    from lanegcn_amd import lanercnn as R
    mods = {"roi": R.LaneRoI(M.config, 128), "ggn": R.GlobalGraphNet(M.config), "pool": R.LanePooling(128, 128),
            "input": R.LaneInput(M.config)}
    for key, m in mods.items():
        mine = [[k, list(v.shape)] for k, v in m.state_dict().items()]
        assert mine == names[key], key


@pytest.mark.gpu
@pytest.mark.parametrize("mma", ["f32", "bf16x3", "f16x2"])
def test_hip_modules_vs_reference_captures(fx, mma):
    g, names = fx
    import lanegcn_amd  # noqa: F401
    from lanegcn_amd import lanegcn as M
    from lanegcn_amd import lanercnn as R
    from lanegcn_amd import ops
    prev = ops.get_mma()
    ops.set_mma(mma)
    try:
        seed = int(g["seed"])
        scenes, li, ctx_g, tgt_g, dev = inputs(g, "cuda")
        mods = {"roi": R.LaneRoI(M.config, 128), "ggn": R.GlobalGraphNet(M.config), "pool": R.LanePooling(128, 128),
                "input": R.LaneInput(M.config)}
        for i, (key, m) in enumerate(mods.items()):
            m.load_state_dict(OR.seeded_state([(k, tuple(s)) for k, s in names[key]], seed + i), strict=True)
            m.cuda().eval()
        x = dev(g["roi/x"])
        with torch.no_grad():
            graph = M.graph_gather([s["graph"] for s in scenes])
            got = mods["roi"](x, graph)
            assert float((got.cpu() - torch.from_numpy(g["roi/out"])).abs().max()) <= FTOL
            got = mods["ggn"](torch.relu(x), graph)
            assert float((got.cpu() - torch.from_numpy(g["ggn/out"])).abs().max()) <= FTOL
            got = mods["input"](li)
            assert float((got.cpu() - torch.from_numpy(g["li/out"])).abs().max()) <= FTOL
            tfeat = dev(g["pool/tfeat"])
            keep = tfeat.clone()
            got = mods["pool"](dev(g["pool/cfeat"]), ctx_g, tfeat, tgt_g, 6.0)
            assert float((got.cpu() - torch.from_numpy(g["pool/out"])).abs().max()) <= FTOL
            assert torch.equal(tfeat, keep)                                  # the input is not modified
            # early return of GlobalGraphNet (lanercnn.py:538-544) and the forward-only modules under autograd
            empty = dict(graph)
            empty["pre"] = graph["pre"][:-1] + [{"u": graph["pre"][-1]["u"][:0], "v": graph["pre"][-1]["v"][:0]}]
            r = mods["ggn"](torch.relu(x), empty)
            assert isinstance(r, tuple) and len(r) == 1 and r[0].numel() == 0
        # LanePooling / LaneInput under autograd: outputs as under no_grad, gradients of inputs and of every parameter
        # against torch autograd through the CPU oracle (fp32) on the same weights
        scenes_c, li_c, ctx_c, tgt_c, _ = inputs(g)
        sd_pool = OR.seeded_state([(k, tuple(s)) for k, s in names["pool"]], seed + 2)
        sd_ref = {"pool." + k: v.clone().requires_grad_(True) for k, v in sd_pool.items()}
        cf_ref, tf_ref = torch.from_numpy(g["pool/cfeat"]).requires_grad_(True), torch.from_numpy(g["pool/tfeat"]).requires_grad_(True)
        wgt = torch.from_numpy(np.random.default_rng(5).normal(0, 1, g["pool/out"].shape).astype(np.float32))
        (OR.lane_pooling(cf_ref, ctx_c, tf_ref, tgt_c, sd_ref)[0] * wgt).sum().backward()
        cf, tf = dev(g["pool/cfeat"]).requires_grad_(True), dev(g["pool/tfeat"]).requires_grad_(True)
        mods["pool"].zero_grad()
        out = mods["pool"](cf, ctx_g, tf, tgt_g, 6.0)
        assert float((out.detach().cpu() - torch.from_numpy(g["pool/out"])).abs().max()) <= FTOL
        (out * wgt.cuda()).sum().backward()
        rel = lambda a, b: float((a.cpu() - b).abs().max()) / max(1e-6, float(b.abs().max()))
        assert rel(cf.grad, cf_ref.grad) <= 2e-4 and rel(tf.grad, tf_ref.grad) <= 2e-4
        for k, prm in mods["pool"].named_parameters():
            assert rel(prm.grad, sd_ref["pool." + k].grad) <= 2e-4, k
        sd_in = OR.seeded_state([(k, tuple(s)) for k, s in names["input"]], seed + 3)
        sd_ref = {"input." + k: v.clone().requires_grad_(True) for k, v in sd_in.items()}
        wgt = torch.from_numpy(np.random.default_rng(6).normal(0, 1, g["li/out"].shape).astype(np.float32))
        (OR.lane_input(li_c, sd_ref) * wgt).sum().backward()
        mods["input"].zero_grad()
        mods["input"].train()
        out = mods["input"](li)
        assert float((out.detach().cpu() - torch.from_numpy(g["li/out"])).abs().max()) <= FTOL
        (out * wgt.cuda()).sum().backward()
        for k, prm in mods["input"].named_parameters():
            assert rel(prm.grad, sd_ref["input." + k].grad) <= 2e-4, k
        # LaneRoI trains through the LaneConv autograd path
        xr = dev(g["roi/x"]).requires_grad_(True)
        out = mods["roi"](xr, M.graph_gather([s["graph"] for s in scenes]))
        out.sum().backward()
        assert xr.grad is not None and torch.isfinite(xr.grad).all()
        assert float((out.detach().cpu() - torch.from_numpy(g["roi/out"])).abs().max()) <= FTOL
    finally:
        ops.set_mma(prev)
