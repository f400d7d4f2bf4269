"""GPU: the weight-stationary LaneConv path (lgcn_lc_plan_build + lgcn_laneconv_fwd, reference lanegcn.py:331-362).

* the device plan is integer work: bit-exact against its numpy restatement (tests/lc_plan_ref.py);
* the layer output against the oracle's index_add_ chain at 1e-4 (north_star) and against the one-launch
  lgcn_agg_mlp implementation of the same arithmetic, for every grouping / capacity (only the fp32 summation
  order of the partial sums changes with them)."""
import numpy as np
import pytest
import torch

import lc_plan_ref as R
from conftest import to_torch_scene
from oracle import lanegcn_oracle as O

pytestmark = pytest.mark.gpu
FTOL = 1e-4


@pytest.fixture(scope="module")
def hip():
    import lanegcn_amd  # noqa: F401
    from lanegcn_amd import lanegcn as M
    from lanegcn_amd import ops
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return M, ops


@pytest.fixture(autouse=True, params=["bf16x3", "f16x2"])
def mma_mode(request, hip):
    _, ops = hip
    prev = ops.get_mma()
    ops.set_mma(request.param)
    yield request.param
    ops.set_mma(prev)


def multigraph(rng, n):
    def edges(m, hot=False):
        u = rng.integers(0, n, m)
        if hot:
            u = np.where(rng.random(m) < 0.5, rng.integers(0, 12, m), u)
        return {"u": torch.from_numpy(u), "v": torch.from_numpy(rng.integers(0, n, m))}

    return {"pre": [edges(m, hot=(i == 0)) for i, m in enumerate((900, 700, 0, 350, 40, 500))],
            "suc": [edges(m) for m in (800, 0, 600, 300, 3, 450)],
            "left": edges(260, hot=True), "right": edges(0)}


def coo(graph):
    us, vs = [], []
    for i in range(len(graph["pre"])):
        for k1 in ("pre", "suc"):
            us.append(graph[k1][i]["u"])
            vs.append(graph[k1][i]["v"])
    for k1 in ("left", "right"):
        us.append(graph[k1]["u"])
        vs.append(graph[k1]["v"])
    return us, vs


def check_plan(ops, lane, n_groups, cap, variant=0):
    lcp = ops.lc_plan(lane, n_groups=n_groups, cap=cap, variant=variant)
    torch.cuda.synchronize()
    M_, n = lcp.rows_per_block, lane.n_nodes
    got = R.split_device_plan(lcp.plan.cpu().numpy(), n, M_, lcp.cap)
    rowptr, col = lane.rowptr.cpu().numpy(), lane.col.cpu().numpy()
    want = R.lc_plan_ref(rowptr, col, n, lane.n_rel, M_, lcp.cap, lcp.gstart)
    assert np.array_equal(got["hdr"], want["hdr"])
    assert np.array_equal(got["mask"], want["mask"])
    written = want["loc"] >= 0
    assert np.array_equal(got["loc"][written], want["loc"][written])
    for slot in np.nonzero(want["hdr"][:, 0] > 0)[0]:
        k = int(want["hdr"][slot, 1])
        assert np.array_equal(got["src"][slot, :k], want["src"][slot, :k]), slot
    assert R.plan_edges(got, col, n, M_) == R.csr_edges(rowptr, col, n, lane.n_rel)
    return lcp


def needs_tiled(ops):
    if ops.lc_config() is None:
        pytest.skip("this matrix mode runs LaneConv on lgcn_agg_mlp (no weight-stationary kernel: round 3)")


def test_plan_bit_exact_multigraph(hip):
    M, ops = hip
    needs_tiled(ops)
    rng = np.random.default_rng(17)
    n = 16 * 23 + 5
    us, vs = coo(multigraph(rng, n))
    lane = ops.csr_build([u.cuda() for u in us], [v.cuda() for v in vs], n)
    for variant in (0, 1, 2):
        m_rows, cap_max = ops.lc_config(variant=variant)
        for n_groups, cap in ((1, None), (4, None), (15, None), (2, m_rows), (1, m_rows + 7)):
            check_plan(ops, lane, n_groups, cap, variant)


def test_plan_bit_exact_synthetic_scenes(hip):
    M, ops = hip
    needs_tiled(ops)
    from lanegcn_amd import data as gen
    scenes = [to_torch_scene(s) for s in gen.synth_batch("S2", seed=3, n_scenes=3)]
    graph = M.graph_gather([s["graph"] for s in scenes])
    lane = M.lane_plan(graph)
    for n_groups, variant in ((1, 0), (4, 1), (1, 1), (1, 2)):
        lcp = check_plan(ops, lane, n_groups, None, variant)
    # chains with dilations and left/right partners: far fewer distinct sources than edges
    hdr = R.split_device_plan(lcp.plan.cpu().numpy(), lane.n_nodes, lcp.rows_per_block, lcp.cap)["hdr"]
    assert hdr[hdr[:, 0] > 0, 1].max() <= lcp.cap


def test_layer_vs_oracle_and_fused_kernel(hip, ref_state_names):
    """M2M (4 LaneConv layers) on a multigraph (in-degree up to 9+, duplicate edges, empty relations, ragged row
    count) for several groupings and forced item splits, against the oracle and the one-launch kernel."""
    M, ops = hip
    rng = np.random.default_rng(17)
    n = 16 * 23 + 5
    sd = O.seeded_state(ref_state_names, 11)
    m2m = M.M2M(M.config)
    m2m.load_state_dict({k[4:]: v for k, v in sd.items() if k.startswith("m2m.")})
    m2m = m2m.cuda().eval()
    graph = multigraph(rng, n)
    feat = torch.from_numpy(rng.normal(0, 1, (n, 128)).astype(np.float32)).relu()
    want = O.m2m(feat, graph, sd).numpy()
    us, vs = coo(graph)
    keys = M.rel_keys(6)

    def layers(lcp):
        part = ops.lc_part(lcp)
        x = feat.cuda()
        for i in range(4):
            wps = [ops.packed(m2m.fuse["ctr"][i].weight)]
            wps += [ops.packed(m2m.fuse[k][i].weight) if lane.n_edges[r] > 0 else None for r, k in enumerate(keys)]
            c2 = m2m.fuse["ctr2"][i]
            x = ops.laneconv_fwd(x, lcp, wps, M._gn(m2m.fuse["norm"][i]), ops.packed(c2.linear.weight),
                                 M._gn(c2.norm), part=part)
        return x.cpu().numpy()

    with torch.no_grad():
        lane = ops.csr_build([u.cuda() for u in us], [v.cuda() for v in vs], n)
        fused = M.lane_conv(m2m.fuse, feat.cuda(), lane, 6, impl="fused").cpu().numpy()
        assert float(np.abs(fused - want).max()) <= FTOL
        got = M.lane_conv(m2m.fuse, feat.cuda(), lane, 6, impl="tiled").cpu().numpy()     # default plan (bf16x3: the one-launch kernel)
        assert float(np.abs(got - want).max()) <= FTOL
        for variant in ((0, 1, 2) if ops.lc_config() is not None else ()):
            m_rows, _ = ops.lc_config(variant=variant)
            # one group = the layer is finished inside the launch; several = partial sums + combine launch;
            # cap = m_rows forces the plan to split groups into several items
            for n_groups, cap in ((1, None), (4, None), (15, None), (3, m_rows), (1, m_rows), (1, m_rows + 7)):
                got = layers(ops.lc_plan(lane, n_groups=n_groups, cap=cap, variant=variant))
                err = float(np.abs(got - want).max())
                assert err <= FTOL, (variant, n_groups, cap, err)
                assert float(np.abs(got - fused).max()) <= 2e-5, (variant, n_groups, cap)


def test_sixteen_wave_short_shape(hip, ref_state_names):
    """lgcn_laneconv_fwd with waves = 16 (K split four ways, the four partial tiles merged in two steps) on the short
    one-group shape: against the oracle and the 8-wave launch, ragged row count, multigraph; refused for other shapes."""
    M, ops = hip
    needs_tiled(ops)
    rng = np.random.default_rng(23)
    n = 48 * 7 + 19
    sd = O.seeded_state(ref_state_names, 11)
    m2m = M.M2M(M.config)
    m2m.load_state_dict({k[4:]: v for k, v in sd.items() if k.startswith("m2m.")})
    m2m = m2m.cuda().eval()
    graph = multigraph(rng, n)
    feat = torch.from_numpy(rng.normal(0, 1, (n, 128)).astype(np.float32)).relu()
    want = O.m2m(feat, graph, sd).numpy()
    us, vs = coo(graph)
    keys = M.rel_keys(6)
    with torch.no_grad():
        lane = ops.csr_build([u.cuda() for u in us], [v.cuda() for v in vs], n)
        lcp = ops.lc_plan(lane, n_groups=1, variant=2)

        def layers(waves, plan=lcp):
            x = feat.cuda()
            for i in range(4):
                wps = [ops.packed(m2m.fuse["ctr"][i].weight)]
                wps += [ops.packed(m2m.fuse[k][i].weight) if lane.n_edges[r] > 0 else None for r, k in enumerate(keys)]
                c2 = m2m.fuse["ctr2"][i]
                x = ops.laneconv_fwd(x, plan, wps, M._gn(m2m.fuse["norm"][i]), ops.packed(c2.linear.weight), M._gn(c2.norm), waves=waves)
            return x.cpu().numpy()

        a8, a16 = layers(8), layers(16)
        assert float(np.abs(a16 - want).max()) <= FTOL
        assert float(np.abs(a16 - a8).max()) <= 2e-5
        assert np.array_equal(a16, layers(16)), "not bitwise repeatable"
        # other shapes / several groups: the request is ignored by ops (8 waves) and refused by the C ABI
        other = ops.lc_plan(lane, n_groups=1, variant=0)
        assert np.array_equal(layers(16, other), layers(8, other))


def test_mapnet_s1_vs_oracle(hip, ref_state_names):
    """BASELINE config 2: MapNet only on the one merged 10,008-node / 59,952-edge graph (S1), vs the oracle."""
    M, ops = hip
    from lanegcn_amd import data as gen
    sd = O.seeded_state(ref_state_names, 7)
    mn = M.MapNet(M.config)
    mn.load_state_dict({k[8:]: v for k, v in sd.items() if k.startswith("map_net.")})
    mn = mn.cuda().eval()
    scenes = [to_torch_scene(s) for s in gen.synth_batch("S1", seed=2)]
    with torch.no_grad():
        graph = M.graph_gather([s["graph"] for s in scenes])
        got, _, _ = mn(graph)
        got2, _, _ = mn(graph)
    assert got.shape[0] == 10008
    want = O.mapnet(O.graph_gather([s["graph"] for s in scenes]), sd)
    err = float((got.cpu() - want).abs().max())
    assert err <= FTOL, err
    assert torch.equal(got, got2), "not bitwise repeatable"


def test_mapnet_s1_exact_f32_mode_vs_oracle(hip, ref_state_names, mma_mode):
    """BASELINE config 2 as stated: MapNet LaneConv only, 10k nodes / 60k edges, fp32 (the exact-f32 MFMA mode)."""
    M, ops = hip
    if mma_mode != "f16x2":
        pytest.skip("runs once")
    from lanegcn_amd import data as gen
    prev = ops.get_mma()
    ops.set_mma("f32")
    try:
        sd = O.seeded_state(ref_state_names, 7)
        mn = M.MapNet(M.config)
        mn.load_state_dict({k[8:]: v for k, v in sd.items() if k.startswith("map_net.")})
        mn = mn.cuda().eval()
        scenes = [to_torch_scene(s) for s in gen.synth_batch("S1", seed=2)]
        with torch.no_grad():
            got, _, _ = mn(M.graph_gather([s["graph"] for s in scenes]))
        want = O.mapnet(O.graph_gather([s["graph"] for s in scenes]), sd)
        assert got.shape == (10008, 128)
        assert float((got.cpu() - want).abs().max()) <= FTOL
    finally:
        ops.set_mma(prev)


def test_bf16_hot_path_s2_vs_oracle(hip, ref_state_names, mma_mode):
    """BASELINE config 3: the full FusionNet path (MapNet + A2M + M2M + M2A + A2A), batch 32 (S2), single bf16
    product.  No reference counterpart exists for bf16; bar = 2e-2 of each stage's feature scale vs the fp32 oracle
    (hard part 6); index work is the same bit-exact kernels as in every mode."""
    M, ops = hip
    if mma_mode != "f16x2":
        pytest.skip("runs once")
    from lanegcn_amd import data as gen
    from test_gpu_parity import make_modules, run_hot_path
    prev = ops.get_mma()
    ops.set_mma("bf16")
    try:
        sd = O.seeded_state(ref_state_names, 3)
        mods = make_modules(M, sd)
        scenes = [to_torch_scene(s) for s in gen.synth_batch("S2", seed=1)]
        A = sum(len(s["ctrs"]) for s in scenes)
        actors = torch.from_numpy(np.random.default_rng(2).normal(0, 1, (A, 128)).astype(np.float32)).relu()
        out, _ = run_hot_path(M, mods, scenes, actors)
        want = O.hot_path(O.graph_gather([s["graph"] for s in scenes]), actors, [s["ctrs"] for s in scenes], sd)
        for k in ("map_net", "a2m", "m2m", "m2a", "a2a"):
            w = want[k].numpy()
            err, scale = float(np.abs(out[k] - w).max()), float(np.abs(w).max())
            assert err <= 2e-2 * scale, (k, err, scale)
    finally:
        ops.set_mma(prev)


def test_bf16_mode_s2_vs_oracle(hip, ref_state_names, mma_mode):
    """BASELINE config 3 (bf16): one bf16 product, S2 batch; bar 2e-2 relative to the feature scale."""
    M, ops = hip
    if mma_mode != "f16x2":
        pytest.skip("runs once")
    from lanegcn_amd import data as gen
    prev = ops.get_mma()
    ops.set_mma("bf16")
    try:
        sd = O.seeded_state(ref_state_names, 3)
        mn = M.MapNet(M.config)
        mn.load_state_dict({k[8:]: v for k, v in sd.items() if k.startswith("map_net.")})
        mn = mn.cuda().eval()
        scenes = [to_torch_scene(s) for s in gen.synth_batch("S2", seed=1)]
        with torch.no_grad():
            got, _, _ = mn(M.graph_gather([s["graph"] for s in scenes]))
        want = O.mapnet(O.graph_gather([s["graph"] for s in scenes]), sd)
        scale = float(want.abs().max())
        err = float((got.cpu() - want).abs().max())
        assert err <= 2e-2 * scale, (err, scale)
    finally:
        ops.set_mma(prev)


def test_f16x2_overflow_is_seen_and_rerouted(hip, ref_state_names, mma_mode):
    """Operands beyond fp16's 65504 in the default f16x2 mode: the forward comes back with NaN rows (never with
    plausible numbers: every ReLU keeps a NaN), the device flag says so, and the guarded entry points re-run in
    bf16x3 -- finite and equal to the oracle -- or raise, by policy."""
    M, ops = hip
    if mma_mode != "f16x2":
        pytest.skip("f16x2 only")
    from lanegcn_amd._lib import LgcnError
    rng = np.random.default_rng(23)
    n = 16 * 23 + 5
    sd = O.seeded_state(ref_state_names, 11)
    m2m = M.M2M(M.config)
    m2m.load_state_dict({k[4:]: v for k, v in sd.items() if k.startswith("m2m.")})
    m2m = m2m.cuda().eval()
    graph = multigraph(rng, n)
    feat = torch.from_numpy(rng.normal(0, 1, (n, 128)).astype(np.float32)).relu()
    feat[7, 3] = 1.0e5                 # one activation beyond fp16's range
    feat[200, :] *= 3.0e4              # a row that sums past it in the gathers
    want = O.m2m(feat, graph, sd)
    assert torch.isfinite(want).all()
    dgraph = {"pre": [{k: v.cuda() for k, v in e.items()} for e in graph["pre"]],
              "suc": [{k: v.cuda() for k, v in e.items()} for e in graph["suc"]],
              "left": {k: v.cuda() for k, v in graph["left"].items()},
              "right": {k: v.cuda() for k, v in graph["right"].items()}, "feats": feat.cuda()}
    scale = float(want.abs().max())
    with torch.no_grad():
        for impl in ("tiled", "fused"):
            ops.set_laneconv_impl(impl)
            try:
                ops.set_guard("off")
                raw = m2m(feat.cuda(), dict(dgraph))
                assert not torch.isfinite(raw).all(), impl + ": the overflow must surface as NaN, not as numbers"
                flag = torch.zeros(1, dtype=torch.int32, device="cuda")
                ops.check_finite(flag, raw)
                assert int(flag.item()) == 1
                ops.set_guard("raise")
                with pytest.raises(LgcnError):
                    m2m(feat.cuda(), dict(dgraph))
                ops.set_guard("reroute")
                got = m2m(feat.cuda(), dict(dgraph)).cpu()
                assert torch.isfinite(got).all()
                assert float((got - want).abs().max()) <= 1e-4 * max(1.0, scale), impl
                assert ops.get_mma() == "f16x2"          # the mode is restored
            finally:
                ops.set_guard("reroute")
                ops.set_laneconv_impl("tiled")
