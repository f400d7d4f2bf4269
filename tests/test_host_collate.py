"""CPU: host-side integer logic -- the flat collate of engine.py against the oracle's graph_gather, and property
tests (hypothesis) of the product's dilated_nbrs against the scipy restatement."""
import numpy as np
import pytest
import torch
from hypothesis import given, settings
from hypothesis import strategies as st

from conftest import to_torch_scene
from oracle import lanegcn_oracle as O


@pytest.fixture(scope="module")
def gen():
    import lanegcn_amd  # noqa: F401
    from lanegcn_amd import data
    return data


def test_collate_flat_reproduces_graph_gather(gen):
    """idx_local + seg_base per segment (what lgcn_graph_gather computes on the device) == the reference's
    per-scene offset adds and cats (lanegcn.py:191-208), for ragged scenes incl. one without left/right edges."""
    from lanegcn_amd.engine import collate_flat
    rng = np.random.default_rng(3)
    scenes = [gen.synth_scene(rng, [4, 6], 7), gen.synth_scene(rng, [5], 3), gen.synth_scene(rng, [4, 4, 4], 11)]
    for k in ("left", "right"):
        scenes[1]["graph"][k] = {"u": np.zeros(0, np.int64), "v": np.zeros(0, np.int64)}
    fb = collate_flat(scenes, device="cpu")
    want = O.graph_gather([to_torch_scene(s)["graph"] for s in scenes])
    seg = np.searchsorted(fb.seg_off.numpy(), np.arange(fb.idx_local.numel()), side="right") - 1
    glob = fb.idx_local.numpy() + fb.seg_base.numpy()[seg]
    r = 0
    for i in range(6):
        for k1 in ("pre", "suc"):
            (ua, ub), (va, vb) = fb.rel_slices[r]
            assert np.array_equal(glob[ua:ub], want[k1][i]["u"].numpy())
            assert np.array_equal(glob[va:vb], want[k1][i]["v"].numpy())
            r += 1
    for k1 in ("left", "right"):
        (ua, ub), (va, vb) = fb.rel_slices[r]
        assert np.array_equal(glob[ua:ub], want[k1]["u"].numpy()) and np.array_equal(glob[va:vb], want[k1]["v"].numpy())
        r += 1
    assert fb.n_nodes == sum(s["graph"]["num_nodes"] for s in scenes) and fb.n_actors == 21
    assert fb.node_off.tolist() == [0, 180, 270, 486] and fb.actor_off.tolist() == [0, 7, 10, 21]
    assert fb.cap_a2m == 180 * 7 + 90 * 3 + 216 * 11 and fb.cap_a2a == 49 + 9 + 121
    assert torch.equal(fb.node_ctrs, torch.cat([torch.from_numpy(s["graph"]["ctrs"]) for s in scenes]))
    assert fb.n_edges == [len(want[k][i]["u"]) for i in range(6) for k in ("pre", "suc")] + [len(want["left"]["u"]), len(want["right"]["u"])]


def test_collate_flat_int16_indices(gen):
    """Preprocessed scenes store indices as int16 (preprocess_data.py:230-238): same flat batch."""
    from lanegcn_amd.engine import collate_flat
    a = gen.synth_batch("S2", seed=4, n_scenes=2)
    b = gen.synth_batch("S2", seed=4, n_scenes=2, idx_dtype=np.int16)
    fa, fbb = collate_flat(a, device="cpu"), collate_flat(b, device="cpu")
    assert fa.idx_local.dtype == torch.int64 and torch.equal(fa.idx_local, fbb.idx_local)


@settings(max_examples=40, deadline=None)
@given(st.integers(2, 40), st.integers(0, 120), st.integers(0, 2 ** 31 - 1))
def test_dilated_nbrs_property(n, m, seed):
    """For random multigraphs (duplicates, self loops, empty): the edge SETS of A^2 .. A^32 equal scipy's, the
    output is sorted by (u, v) and duplicate free."""
    import lanegcn_amd  # noqa: F401
    from lanegcn_amd import data
    rng = np.random.default_rng(seed)
    u, v = rng.integers(0, n, m), rng.integers(0, n, m)
    mine = data.dilated_nbrs({"u": u, "v": v}, n, 6)
    ref = O.dilated_nbrs({"u": u, "v": v}, n, 6)
    assert len(mine) == 5
    for a, b in zip(mine, ref):
        ka, kb = a["u"] * n + a["v"], np.unique(b["u"] * n + b["v"])
        assert np.array_equal(ka, kb)            # sorted + unique + same set
        assert a["u"].dtype == np.int64 and a["v"].dtype == np.int64


def test_pair_search_oracle_modes():
    """legacy vs fixed offsets differ exactly by the skipped scenes (lanegcn.py:681-687)."""
    a = [torch.zeros(3, 2), torch.zeros(2, 2) + 1000.0, torch.zeros(4, 2)]
    c = [torch.zeros(2, 2), torch.zeros(5, 2), torch.zeros(1, 2)]
    hi_l, wi_l = O.pair_search(a, c, 1.0, legacy=True)
    hi_f, wi_f = O.pair_search(a, c, 1.0, legacy=False)
    assert hi_l.tolist() == [0, 0, 1, 1, 2, 2, 3, 4, 5, 6] and wi_l.tolist() == [0, 1, 0, 1, 0, 1, 2, 2, 2, 2]
    assert hi_f.tolist() == [0, 0, 1, 1, 2, 2, 5, 6, 7, 8] and wi_f.tolist() == [0, 1, 0, 1, 0, 1, 7, 7, 7, 7]


def test_flat_scene_file_roundtrip(gen, tmp_path):
    """flatfile.write_scenes / FlatSceneFile.batch (row f2) cut the same FlatBatch as collate_flat does from the
    scene dicts, for an arbitrary subset and order of scenes, without pickles."""
    from lanegcn_amd.engine import FullNetEngine, collate_flat
    from lanegcn_amd.flatfile import FlatSceneFile, write_scenes
    rng = np.random.default_rng(8)
    scenes = [gen.synth_scene(rng, [4, 5], 6), gen.synth_scene(rng, [6], 3), gen.synth_scene(rng, [4, 4, 4], 9),
              gen.synth_scene(rng, [5], 2)]
    scenes[1]["graph"]["left"] = {"u": np.zeros(0, np.int64), "v": np.zeros(0, np.int64)}
    path = str(tmp_path / "split.npz")
    write_scenes(path, scenes)
    store = FlatSceneFile(path)
    assert len(store) == 4
    pick = [2, 0, 3]
    fb, ex = store.batch(pick, device="cpu")
    want = collate_flat([scenes[i] for i in pick], device="cpu")
    for name in ("n_scenes", "n_nodes", "n_actors", "num_scales", "rel_slices", "cap_a2m", "cap_a2a", "n_edges"):
        assert getattr(fb, name) == getattr(want, name), name
    for name in ("node_ctrs", "node_feats", "turn", "control", "intersect", "actor_ctrs", "node_off", "actor_off",
                 "idx_local", "seg_off", "seg_base"):
        assert torch.equal(getattr(fb, name), getattr(want, name)), name
    feats, rot, orig = FullNetEngine.actor_inputs([scenes[i] for i in pick], device="cpu")
    assert torch.equal(ex["actor_feats"], feats) and torch.equal(ex["rot"], rot) and torch.equal(ex["orig"], orig)
    assert ex["sizes"] == [9, 6, 2] and ex["gt_preds"].shape == (17, 30, 2) and ex["has_preds"].dtype == torch.bool
