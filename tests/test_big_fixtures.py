"""Rows f3 / f4 AT SIZE (>= 1 k nodes per scene): tests/golden/big_f3f4.npz holds what the
reference's own preprocess / data.dilated_nbrs / lanercnn modules returned for the inputs of
tests/golden/make_golden.py:big_inputs() (regenerated here from their seed).  Integer outputs are stored in full or as
SHA-1 of the sorted edge set, features as every 16th row + float64 column sums.
CPU: the oracle restatements against the fixture.  GPU: the product (lgcn_cross_edges, lgcn_bool_square*, the lanercnn
modules on the HIP kernels) against it."""
import hashlib
import importlib.util
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR, to_torch_scene
from oracle import graphgen_oracle as GO
from oracle import lanegcn_oracle as O
from oracle import lanercnn_oracle as OR

FTOL = 1e-4


@pytest.fixture(scope="module")
def big():
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(GOLDEN_DIR, "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)                      # defines functions only; the reference is not imported
    with np.load(os.path.join(GOLDEN_DIR, "big_f3f4.npz")) as z:
        fx = {k: z[k] for k in z.files}
    return mg, fx, mg.big_inputs()


def sha(a):
    return np.frombuffer(hashlib.sha1(np.ascontiguousarray(a).tobytes()).digest(), np.uint8)


def edges_sha(u, v):
    uv = np.stack([np.asarray(u, np.int64), np.asarray(v, np.int64)], 1)
    uv = uv[np.lexsort((uv[:, 1], uv[:, 0]))]
    return sha(uv), len(uv)


def check_feat(fx, key, got):
    got = got.detach().cpu().numpy()
    assert float(np.abs(got[::16] - fx[key + "/rows16"]).max()) <= FTOL, key
    cs = got.astype(np.float64).sum(0)
    assert float(np.abs(cs - fx[key + "/colsum"]).max()) <= FTOL * got.shape[0], key      # every row contributes <= FTOL


def pool_graphs(scenes, tgt, dev):
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    graphs = [s["graph"] for s in scenes]
    ctx_g = {"ctrs": [t(g["ctrs"].astype(np.float32)) for g in graphs],
             "pose": [t(np.concatenate([g["ctrs"], g["feats"]], 1).astype(np.float32)) for g in graphs]}
    tgt_g = {"ctrs": [t(c) for c, _ in tgt], "pose": [t(p) for _, p in tgt]}
    return ctx_g, tgt_g


def test_oracles_vs_reference_at_size(big):
    mg, fx, (scenes, x, cfeat, tgt, tfeat, raw) = big
    for i, g in enumerate(raw):
        assert len(g["lane_idcs"]) >= 1000
        res = GO.preprocess(g, 6.0)
        for side in ("left", "right"):
            assert np.array_equal(res[side]["u"], fx["pp%d/%s/u" % (i, side)]) and np.array_equal(res[side]["v"], fx["pp%d/%s/v" % (i, side)])
    import lanegcn_amd  # noqa: F401
    from lanegcn_amd import data as gen
    gr = scenes[0]["graph"]
    for k1 in ("pre", "suc"):
        nb = gen.dilated_nbrs({"u": gr[k1][0]["u"], "v": gr[k1][0]["v"]}, int(gr["num_nodes"]), 6)
        for j, e in enumerate(nb):
            h, n = edges_sha(e["u"], e["v"])
            assert n == int(fx["dil/%s/%d/n" % (k1, j + 1)]) and np.array_equal(h, fx["dil/%s/%d/sha1" % (k1, j + 1)]), (k1, j)
    seed = int(fx["seed"])
    names = __import__("json").load(open(os.path.join(GOLDEN_DIR, "lanercnn_state_names.json")))
    ts = [to_torch_scene(s) for s in scenes]
    graph = O.graph_gather([s["graph"] for s in ts])
    sd = lambda key, i: {key + "." + k: v for k, v in OR.seeded_state([(k, tuple(s)) for k, s in names[key]], seed + i).items()}
    check_feat(fx, "roi", OR.lane_roi(torch.from_numpy(x), graph, sd("roi", 0)))
    check_feat(fx, "ggn", OR.global_graph_net(torch.relu(torch.from_numpy(x)), graph, sd("ggn", 1)))
    ctx_g, tgt_g = pool_graphs(scenes, tgt, "cpu")
    out, hi, wi = OR.lane_pooling(torch.from_numpy(cfeat), ctx_g, torch.from_numpy(tfeat), tgt_g, sd("pool", 2))
    assert len(wi) == int(fx["pool/n_pairs"]) and np.array_equal(sha(wi.numpy().astype(np.int64)), fx["pool/wi_sha1"])
    check_feat(fx, "pool", out)


@pytest.mark.gpu
@pytest.mark.parametrize("mma", ["f32", "bf16x3", "f16x2"])
def test_hip_f3_f4_vs_reference_at_size(big, mma):
    mg, fx, (scenes, x, cfeat, tgt, tfeat, raw) = big
    import lanegcn_amd  # noqa: F401
    from lanegcn_amd import lanegcn as M
    from lanegcn_amd import lanercnn as R
    from lanegcn_amd import ops, preprocess_data as P
    prev = ops.get_mma()
    ops.set_mma(mma)
    try:
        if mma == "f32":      # integer work: once
            for i, g in enumerate(raw):
                tg = {k: torch.from_numpy(v).cuda() for k, v in g.items()}
                tg["idx"] = i
                res = P.preprocess(tg, 6.0)
                for side in ("left", "right"):
                    assert np.array_equal(res[side]["u"], fx["pp%d/%s/u" % (i, side)]), (i, side)
                    assert np.array_equal(res[side]["v"], fx["pp%d/%s/v" % (i, side)]), (i, side)
            gr = scenes[0]["graph"]
            for k1 in ("pre", "suc"):
                e0 = {k: torch.from_numpy(np.asarray(v, np.int64)).cuda() for k, v in gr[k1][0].items()}
                nb = P.dilated_nbrs(e0, int(gr["num_nodes"]), 6)
                for j, e in enumerate(nb):
                    h, n = edges_sha(e["u"].cpu().numpy(), e["v"].cpu().numpy())
                    assert n == int(fx["dil/%s/%d/n" % (k1, j + 1)]) and np.array_equal(h, fx["dil/%s/%d/sha1" % (k1, j + 1)]), (k1, j)
        seed = int(fx["seed"])
        names = __import__("json").load(open(os.path.join(GOLDEN_DIR, "lanercnn_state_names.json")))
        mods = {"roi": R.LaneRoI(M.config, 128), "ggn": R.GlobalGraphNet(M.config), "pool": R.LanePooling(128, 128)}
        for i, (key, m) in enumerate(mods.items()):
            m.load_state_dict(OR.seeded_state([(k, tuple(s)) for k, s in names[key]], seed + i), strict=True)
            m.cuda().eval()
        with torch.no_grad():
            graph = M.graph_gather([to_torch_scene(s)["graph"] for s in scenes])
            xd = torch.from_numpy(x).cuda()
            check_feat(fx, "roi", mods["roi"](xd, graph))
            check_feat(fx, "ggn", mods["ggn"](torch.relu(xd), graph))
            ctx_g, tgt_g = pool_graphs(scenes, tgt, "cuda")
            check_feat(fx, "pool", mods["pool"](torch.from_numpy(cfeat).cuda(), ctx_g, torch.from_numpy(tfeat).cuda(), tgt_g, 6.0))
    finally:
        ops.set_mma(prev)
