"""GPU: graph construction on the device against the reference's own outputs."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR
from golden_io import load_scenes

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hipmods():
    import lanegcn_amd  # noqa: F401
    from lanegcn_amd import ops, preprocess_data
    return ops, preprocess_data


def edge_set(d):
    uv = np.stack([np.asarray(torch.as_tensor(d["u"]).cpu(), np.int64), np.asarray(torch.as_tensor(d["v"]).cpu(), np.int64)], 1)
    return uv[np.lexsort((uv[:, 1], uv[:, 0]))]


def test_preprocess_left_right_bit_exact(hipmods):
    """lgcn_cross_edges behind the reference's `preprocess(graph, cross_dist)` signature: the int16 left / right index
    arrays the reference returned for the six fixture topologies, element for element (ties, the empty-side branch,
    a scene without any edge within 6 m, a one-lane scene)."""
    ops, P = hipmods
    with np.load(os.path.join(GOLDEN_DIR, "graphgen_b6.npz")) as z:
        z = {k: z[k] for k in z.files}
    for i in range(int(z["n_scenes"])):
        g = {k: torch.from_numpy(z["g%d/%s" % (i, k)]).cuda()
             for k in ("ctrs", "feats", "lane_idcs", "pre_pairs", "suc_pairs", "left_pairs", "right_pairs")}
        g["idx"] = i
        out = P.preprocess(g, float(z["cross_dist"]))
        assert out["idx"] == i
        for side in ("left", "right"):
            assert out[side]["u"].dtype == np.int16 and out[side]["v"].dtype == np.int16
            assert np.array_equal(out[side]["u"], z["g%d/%s/u" % (i, side)]), (i, side)
            assert np.array_equal(out[side]["v"], z["g%d/%s/v" % (i, side)]), (i, side)
    with pytest.raises(NameError):      # the reference's cross_angle branch dies on an undefined global `config`: same here
        P.preprocess(g, 6.0, cross_angle=0.5)
    with pytest.raises(Exception):
        P.preprocess({k: (v.cpu() if torch.is_tensor(v) else v) for k, v in g.items()}, 6.0)


def test_preprocess_random_topologies_vs_oracle(hipmods):
    """Random geometry and random lane pairs (most rows fully masked, duplicated pairs, equal distances): against
    the oracle; rows whose heading difference is within 1e-5 of pi / 4 are excluded from the comparison."""
    ops, P = hipmods
    from oracle import graphgen_oracle as GO
    rng = np.random.default_rng(8)
    for trial in range(4):
        n, nl = int(rng.integers(40, 400)), int(rng.integers(3, 30))
        lane = np.sort(rng.integers(0, nl, n))
        lane[-1] = nl - 1
        g = dict(ctrs=np.round(rng.normal(0, 8, (n, 2)), 1).astype(np.float32),      # coarse grid: equal distances occur
                 feats=rng.normal(0, 1, (n, 2)).astype(np.float32), lane_idcs=lane.astype(np.int64),
                 pre_pairs=rng.integers(0, nl, (int(rng.integers(0, 40)), 2)).astype(np.int64),
                 suc_pairs=rng.integers(0, nl, (int(rng.integers(0, 40)), 2)).astype(np.int64),
                 left_pairs=rng.integers(0, nl, (int(rng.integers(1, 30)), 2)).astype(np.int64),
                 right_pairs=rng.integers(0, nl, (int(rng.integers(1, 30)), 2)).astype(np.int64))
        want = GO.preprocess(g, 6.0)
        tg = {k: torch.from_numpy(v).cuda() for k, v in g.items()}
        tg["idx"] = trial
        got = P.preprocess(tg, 6.0)
        for side in ("left", "right"):
            a = dict(zip(got[side]["u"].tolist(), got[side]["v"].tolist()))
            b = dict(zip(want[side]["u"].tolist(), want[side]["v"].tolist()))
            for u in set(a) | set(b):
                if u in a and u in b:
                    assert a[u] == b[u], (trial, side, u)
                else:      # present on one side only: must be a heading-threshold row
                    v = a.get(u, b.get(u))
                    assert GO.heading_margin(g["feats"], np.array([u]), np.array([v]))[0] < 1e-5, (trial, side, u)


def test_dilated_nbrs_on_the_device(hipmods, golden):
    """lgcn_bool_square*: scales 1..5 of the fixture scenes' pre / suc relations against the edge sets the reference's
    own data.dilated_nbrs produced (dil/*), and a random multigraph (duplicate edges, self loops, in-degree up to
    ~10) against the host implementation."""
    ops, P = hipmods
    from lanegcn_amd import data as gen
    scenes = load_scenes(golden)
    seen = 0
    for i, sc in enumerate(scenes):
        n = int(sc["graph"]["num_nodes"])
        for k1 in ("pre", "suc"):
            e0 = {k: torch.from_numpy(np.asarray(v, np.int64)).cuda() for k, v in sc["graph"][k1][0].items()}
            got = P.dilated_nbrs(e0, n, 6)
            assert len(got) == 5
            for j in range(5):
                assert got[j]["u"].dtype == torch.int64 and got[j]["u"].is_cuda
                assert np.array_equal(edge_set(got[j]), golden["dil/%d/%s/%d" % (i, k1, j + 1)]), (i, k1, j + 1)
                seen += 1
    assert seen == 40
    rng = np.random.default_rng(12)
    n, m = 203, 700
    u, v = rng.integers(0, n, m), rng.integers(0, n, m)
    want = gen.dilated_nbrs({"u": u, "v": v}, n, 4)
    got = P.dilated_nbrs({"u": torch.from_numpy(u).cuda(), "v": torch.from_numpy(v).cuda()}, n, 4)
    for a, b in zip(got, want):
        assert np.array_equal(edge_set(a), edge_set(b))
    with pytest.raises(Exception):
        P.dilated_nbrs({"u": torch.from_numpy(u), "v": torch.from_numpy(v)}, n, 4)
