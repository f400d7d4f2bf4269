"""CPU: the plugin surface the reference's drivers bind to (train.py:63-64, test.py:57) -- names, config
keys, state_dict layout, collate format, optimizer schedule -- without touching the GPU."""
import inspect

import numpy as np
import pytest
import torch


@pytest.fixture(scope="module")
def M():
    import lanegcn_amd  # noqa: F401
    from lanegcn_amd import lanegcn
    return lanegcn


def test_state_dict_is_the_reference_layout(M, ref_state_names):
    net = M.Net(M.config)
    mine = [(k, tuple(v.shape)) for k, v in net.state_dict().items()]
    assert mine == ref_state_names                      # 405 tensors, same names, shapes AND order
    assert sum(p.numel() for p in net.parameters()) == 3701161


def test_module_signatures(M):
    sig = lambda f: list(inspect.signature(f).parameters)
    assert sig(M.MapNet.forward) == ["self", "graph"]
    assert sig(M.A2M.forward) == ["self", "feat", "graph", "actors", "actor_idcs", "actor_ctrs"]
    assert sig(M.M2M.forward) == ["self", "feat", "graph"]
    assert sig(M.M2A.forward) == ["self", "actors", "actor_idcs", "actor_ctrs", "nodes", "node_idcs", "node_ctrs"]
    assert sig(M.A2A.forward) == ["self", "actors", "actor_idcs", "actor_ctrs"]
    assert sig(M.Att.forward)[:8] == ["self", "agts", "agt_idcs", "agt_ctrs", "ctx", "ctx_idcs", "ctx_ctrs", "dist_th"]
    assert sig(M.Att.__init__) == ["self", "n_agt", "n_ctx"]
    assert sig(M.graph_gather) == ["graphs"] and sig(M.actor_gather) == ["actors"]
    assert sig(M.Linear.__init__) == ["self", "n_in", "n_out", "norm", "ng", "act"]
    assert callable(M.get_model)


def test_config_keys(M):
    for k, v in dict(n_map=128, n_actor=128, num_scales=6, actor2map_dist=7.0, map2actor_dist=6.0,
                     actor2actor_dist=100.0, batch_size=32, num_mods=6, num_preds=30, opt="adam").items():
        assert M.config[k] == v
    lr = M.config["lr_func"]
    assert lr(0.0) == 1e-3 and lr(31.99) == 1e-3 and lr(32.0) == 1e-4


def test_collate_and_actor_gather(M):
    from lanegcn_amd import data as gen
    scenes = gen.synth_batch("S2", seed=0, n_scenes=3)
    batch = gen.collate_fn(scenes)
    assert set(batch) >= {"feats", "ctrs", "graph", "rot", "orig", "gt_preds", "has_preds"}
    assert all(isinstance(v, list) and len(v) == 3 for v in batch.values())
    assert torch.is_tensor(batch["graph"][0]["pre"][5]["u"])
    # the packed copy for the device rides along as an attribute, not as a key
    hfb, (tracks, rot, orig, sizes) = batch.flat
    assert sizes == [50, 50, 50] and tracks.shape == (150, 3, 20) and hfb.meta["n_nodes"] == 972
    feats, idcs = M.actor_gather(batch["feats"])
    assert feats.shape == (150, 3, 20)
    assert [x.tolist() for x in idcs] == [list(range(50 * i, 50 * i + 50)) for i in range(3)]


def test_to_long_and_optimizer():
    import lanegcn_amd  # noqa: F401
    from lanegcn_amd.utils import Optimizer, StepLR, to_long
    d = {"a": [torch.zeros(3, dtype=torch.int16)], "b": torch.zeros(2)}
    out = to_long(d)
    assert out["a"][0].dtype == torch.int64 and out["b"].dtype == torch.float32
    w = torch.nn.Parameter(torch.ones(4))
    opt = Optimizer([w], {"opt": "adam", "lr_func": StepLR([1e-3, 1e-4], [32])})
    w.grad = torch.ones(4)
    assert opt.step(0.5) == 1e-3 and opt.step(33.0) == 1e-4
    assert opt.opt.param_groups[0]["lr"] == 1e-4


def test_synthetic_workload_sizes():
    """The canonical workloads of SURVEY.md 8(d)."""
    from lanegcn_amd import data as gen
    s2 = gen.synth_batch("S2", seed=0)
    n = sum(s["graph"]["num_nodes"] for s in s2)
    e = sum(sum(len(d["u"]) for d in s["graph"]["pre"] + s["graph"]["suc"]) + 2 * len(s["graph"]["left"]["u"]) for s in s2)
    assert (len(s2), n, e, sum(len(s["ctrs"]) for s in s2)) == (32, 10368, 110592, 1600)
    g = gen.synth_batch("S1", seed=0)[0]["graph"]
    e1 = sum(len(d["u"]) for d in g["pre"] + g["suc"]) + 2 * len(g["left"]["u"])
    assert (g["num_nodes"], e1) == (10008, 59952)
    assert gen.synth_batch("S0", seed=0)[0]["graph"]["num_nodes"] == 648


def test_upsample2_linear_is_interpolate(M):
    import torch.nn.functional as F
    for shape in ((7, 5, 10), (3, 4, 5), (2, 1, 1)):
        x = torch.randn(*shape)
        want = F.interpolate(x, scale_factor=2, mode="linear", align_corners=False)
        assert torch.allclose(M.upsample2_linear(x), want, atol=1e-6)
