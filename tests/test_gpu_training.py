"""GPU: one training step through the drop-in Net (HIP forward + HIP backward of the hot path) against the
reference's own loss, gradients and Adam update (tests/golden/train_b4.npz, captured by make_golden.py)."""
import json
import os

import numpy as np
import pytest
import torch

from golden_io import load_scenes
from oracle import lanegcn_oracle as O

pytestmark = pytest.mark.gpu
GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def train_golden():
    with np.load(os.path.join(GOLDEN_DIR, "train_b4.npz")) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(params=["f32", "f16x2"])
def mma(request):
    import lanegcn_amd  # noqa: F401
    from lanegcn_amd import ops
    prev = ops.get_mma()
    ops.set_mma(request.param)
    yield request.param
    ops.set_mma(prev)


def rel_err(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12))


def test_training_step_matches_reference(golden, train_golden, ref_state_names, mma):
    import lanegcn_amd  # noqa: F401
    from lanegcn_amd import data as gen
    from lanegcn_amd import lanegcn as M
    scenes = load_scenes(golden)
    net = M.Net(M.config)
    net.load_state_dict(O.seeded_state(ref_state_names, int(train_golden["seed"])), strict=True)
    net = net.cuda().train()
    loss_fn = M.Loss(M.config).cuda()
    batch = gen.collate_fn(scenes)
    out = net(batch)
    loss_out = loss_fn(out, batch)
    loss_out["loss"].backward()
    torch.cuda.synchronize()

    # loss (lanegcn.py:740-821)
    assert loss_out["num_cls"] == int(train_golden["loss/num_cls"])
    assert loss_out["num_reg"] == int(train_golden["loss/num_reg"])
    for k in ("cls_loss", "reg_loss", "loss"):
        assert float(loss_out[k].detach()) == pytest.approx(float(train_golden["loss/" + k]), rel=2e-5), k

    # every parameter receives a gradient of the reference's magnitude
    names = json.load(open(os.path.join(GOLDEN_DIR, "param_names.json")))
    params = dict(net.named_parameters())
    assert list(params) == names
    norms = np.array([float(params[n].grad.norm()) if params[n].grad is not None else -1.0 for n in names])
    ref_norms = train_golden["grad_norms"]
    assert (norms >= 0).all()
    bad = [(n, a, b) for n, a, b in zip(names, norms, ref_norms) if abs(a - b) > 2e-3 * b + 1e-6]
    assert not bad, bad[:5]

    # selected gradients element-wise (max error relative to the tensor's largest entry).  Gradients are only
    # piecewise smooth: in f32 mode ONE of the 62,208 ReLU inputs at the A2M.meta output of this fixture has a
    # reference pre-activation of 4.6e-7 and lands on the other side of zero (tools/debug_mask.py), which moves a
    # few upstream entries by ~3e-3 of the tensor scale; the split modes happen not to flip it (all <= 4e-6).
    worst = {}
    for key, ref in train_golden.items():
        if key.startswith("grad/"):
            n = key[5:]
            worst[n] = rel_err(params[n].grad.cpu().numpy(), ref)
    # The looser bar is only granted when the forward really has a ReLU input on the knife edge: count the sign
    # flips of the A2M.meta output (62,208 ReLU inputs) between this build and the oracle, and check that every
    # flipped entry is one whose value is ~0 on both sides.
    flips = meta_relu_flips(net, scenes, ref_state_names, int(train_golden["seed"]))
    if mma == "f32":
        assert 0 < len(flips) <= 2, flips
    else:
        assert len(flips) == 0, flips
    assert all(max(abs(a), abs(b)) <= 1e-5 for a, b in flips), flips
    bar = 5e-3 if flips else 1e-4
    assert max(worst.values()) <= bar, sorted(worst.items(), key=lambda kv: -kv[1])[:5]

    # Adam step through the Optimizer wrapper (utils.py:98-162): lr schedule + update
    opt = M.Optimizer(net.parameters(), M.config)
    assert opt.step(0.0) == float(train_golden["lr"])
    for key, ref in train_golden.items():
        if key.startswith("after/"):
            got = params[key[6:]].detach().cpu().numpy()
            close = np.isclose(got, ref, atol=2e-5, rtol=1e-4)
            # Adam's first step is lr * g / (|g| + 1e-8): entries whose gradient is ~1e-8 amplify the f32-mode
            # mask flip above into a full +-lr move, so there a handful of entries may differ
            assert close.all() if mma != "f32" else close.mean() >= 0.995, (key, float(close.mean()))


def meta_relu_flips(net, scenes, ref_state_names, seed):
    """(value here, value in the oracle) of every A2M.meta output entry (lanegcn.py:387-395) that is zero on one
    side and positive on the other."""
    import lanegcn_amd  # noqa: F401
    from lanegcn_amd import lanegcn as M
    sd = O.seeded_state(ref_state_names, seed)
    tscenes = [{k: v for k, v in s.items()} for s in scenes]
    from conftest import to_torch_scene
    tscenes = [to_torch_scene(s) for s in scenes]
    g_ref = O.graph_gather([s["graph"] for s in tscenes])
    nodes_ref = O.mapnet(g_ref, sd)
    meta_in = torch.cat((nodes_ref, g_ref["turn"], g_ref["control"].unsqueeze(1), g_ref["intersect"].unsqueeze(1)), 1)
    want = O.linear_block(meta_in, sd, "a2m.meta")
    with torch.no_grad():
        graph = M.graph_gather([s["graph"] for s in tscenes])
        was = net.training
        net.eval()
        nodes, _, _ = net.map_net(graph)
        got = net.a2m.fuse_meta(nodes, graph["turn"], graph["control"], graph["intersect"]).cpu()
        net.train(was)
    idx = torch.nonzero((got > 0) != (want > 0))
    return [(float(got[i, j]), float(want[i, j])) for i, j in idx.tolist()]


def stage_relu_flips(net, scenes, ref_state_names, seed):
    """ReLU sign flips at the six stage outputs of the hot path (MapNet, A2M.meta, A2M, M2M, M2A, A2A -- each ends in
    a ReLU) between this build's forward and the oracle's, both fed THIS build's ActorNet output: per stage the number
    of entries that are zero on one side and positive on the other, and the largest magnitude among them."""
    import lanegcn_amd  # noqa: F401
    from lanegcn_amd import lanegcn as M
    from conftest import to_torch_scene
    sd = O.seeded_state(ref_state_names, seed)
    tscenes = [to_torch_scene(s) for s in scenes]
    out = {}
    with torch.no_grad():
        was = net.training
        net.eval()
        actors, idcs = M.actor_gather([s["feats"].cuda() for s in tscenes])
        ctrs = [s["ctrs"].cuda() for s in tscenes]
        a_in = net.actor_net(actors)
        graph = M.graph_gather([s["graph"] for s in tscenes])
        got = {}
        nodes, nidcs, nctrs = net.map_net(graph)
        got["map_net"] = nodes
        got["meta"] = net.a2m.fuse_meta(nodes, graph["turn"], graph["control"], graph["intersect"])
        got["a2m"] = net.a2m(nodes, graph, a_in, idcs, ctrs)
        got["m2m"] = net.m2m(got["a2m"], graph)
        got["m2a"] = net.m2a(a_in, idcs, ctrs, got["m2m"], nidcs, nctrs)
        got["a2a"] = net.a2a(got["m2a"], idcs, ctrs)
        net.train(was)
        g_ref = O.graph_gather([s["graph"] for s in tscenes])
        want = O.hot_path(g_ref, a_in.cpu(), [s["ctrs"] for s in tscenes], sd)
        meta_in = torch.cat((want["map_net"], g_ref["turn"], g_ref["control"].unsqueeze(1), g_ref["intersect"].unsqueeze(1)), 1)
        want["meta"] = O.linear_block(meta_in, sd, "a2m.meta")
    for k in ("map_net", "meta", "a2m", "m2m", "m2a", "a2a"):
        a, b = got[k].cpu(), want[k]
        f = (a > 0) != (b > 0)
        out[k] = (int(f.sum()), float(torch.maximum(a[f].abs().max(), b[f].abs().max())) if f.any() else 0.0,
                  float((a - b).abs().max()))
    return out


@pytest.mark.parametrize("mode", ["f32", "f16x2"])
def test_training_step_batch32_matches_reference(ref_state_names, mode):
    """BASELINE config 4 (fp32; also in the default f16x2 mode): the training step at batch 32 (workload S2, 10,368
    nodes / 1,600 actors) against the reference's own loss and gradients (tests/golden/train_b32.npz, make_golden.py):
    loss scalars, every parameter's gradient norm, and the fixture's gradients element-wise (matrices: every 8th row).
    The gradient bars are tied to EVIDENCE gathered in the same run: the ReLU sign flips at the stage outputs (a flip
    is a pre-activation within rounding of zero that lands on the other side under a different summation order; each
    moves a handful of upstream gradient entries by ~1e-3 of the tensor's scale), counted against the oracle."""
    import lanegcn_amd  # noqa: F401
    from lanegcn_amd import data as gen
    from lanegcn_amd import lanegcn as M
    from lanegcn_amd import ops
    with np.load(os.path.join(GOLDEN_DIR, "train_b32.npz")) as z:
        tg = {k: z[k] for k in z.files}
    prev, prev_det, prev_bench = ops.get_mma(), torch.backends.cudnn.deterministic, torch.backends.cudnn.benchmark
    ops.set_mma(mode)
    torch.backends.cudnn.deterministic, torch.backends.cudnn.benchmark = True, False      # the stock ops' solver choice pinned
    try:
        scenes = gen.synth_batch("S2", seed=int(tg["batch_seed"]))
        assert len(scenes) == 32
        net = M.Net(M.config)
        net.load_state_dict(O.seeded_state(ref_state_names, int(tg["seed"])), strict=True)
        net = net.cuda().train()
        batch = gen.collate_fn([gen.from_numpy(s) for s in scenes])
        loss_out = M.Loss(M.config).cuda()(net(batch), batch)
        loss_out["loss"].backward()
        torch.cuda.synchronize()
        assert loss_out["num_cls"] == int(tg["loss/num_cls"]) and loss_out["num_reg"] == int(tg["loss/num_reg"])
        for k in ("cls_loss", "reg_loss", "loss"):
            assert float(loss_out[k].detach()) == pytest.approx(float(tg["loss/" + k]), rel=5e-5), k
        names = json.load(open(os.path.join(GOLDEN_DIR, "param_names.json")))
        params = dict(net.named_parameters())
        norms = np.array([float(params[n].grad.norm()) if params[n].grad is not None else -1.0 for n in names])
        ref_norms = tg["grad_norms"]
        bad = [(n, a, b) for n, a, b in zip(names, norms, ref_norms) if abs(a - b) > 3e-3 * b + 1e-6]
        assert not bad, bad[:5]
        worst = {}
        for key, ref in tg.items():
            if key.startswith("grad/"):
                g = params[key[5:]].grad.cpu().numpy()
                g = g[::8] if g.shape != ref.shape else g
                assert g.shape == ref.shape, key
                worst[key[5:]] = rel_err(g, ref)
        flips = stage_relu_flips(net, [gen.from_numpy(s) for s in scenes], ref_state_names, int(tg["seed"]))
        n_flips = sum(v[0] for v in flips.values())
        hot = {k: v for k, v in worst.items() if k.split(".")[0] in ("map_net", "a2m", "m2m", "m2a", "a2a")}
        rest = {k: v for k, v in worst.items() if k not in hot}
        print("\n[batch-32 step, %s] flips per stage (count, largest |value|, max |delta| of the stage): %s" % (mode, flips))
        print("[batch-32 step, %s] hot-path gradients: max %.2e median %.2e (%d tensors); other: max %.2e (%d tensors)" %
              (mode, max(hot.values()), float(np.median(list(hot.values()))), len(hot), max(rest.values()) if rest else 0.0, len(rest)))
        # every flipped entry sits on the knife edge (|value| at rounding level on both sides), and the forward agrees
        assert all(v[1] <= 2e-4 for v in flips.values()), flips
        assert all(v[2] <= 1e-4 for v in flips.values()), flips
        if n_flips == 0:
            assert max(hot.values()) <= 1e-4, sorted(hot.items(), key=lambda kv: -kv[1])[:5]
        else:
            # the stage outputs are 6 of the ~45 ReLU layers of the hot path (8 LaneConv layers x 2, 6 Att x 4, stems):
            # flips inside the stages go with the counted ones.  Each moves a few entries by ~1e-3 of a tensor's scale.
            assert max(hot.values()) <= 5e-3, sorted(hot.items(), key=lambda kv: -kv[1])[:5]
            assert np.median(list(hot.values())) <= 1e-4 * (1 + n_flips), (n_flips, sorted(hot.items(), key=lambda kv: -kv[1])[:5])
        assert max(worst.values()) <= 5e-3, sorted(worst.items(), key=lambda kv: -kv[1])[:5]
    finally:
        ops.set_mma(prev)
        torch.backends.cudnn.deterministic, torch.backends.cudnn.benchmark = prev_det, prev_bench


def test_training_forward_equals_inference_forward(golden, ref_state_names, mma):
    """The differentiable composition (grad enabled) and the fused inference kernels compute the same features."""
    import lanegcn_amd  # noqa: F401
    from lanegcn_amd import data as gen
    from lanegcn_amd import lanegcn as M
    scenes = load_scenes(golden)
    net = M.Net(M.config)
    net.load_state_dict(O.seeded_state(ref_state_names, int(golden["seed"])), strict=True)
    net = net.cuda()
    batch = gen.collate_fn(scenes)
    with torch.no_grad():
        ref = net(batch)
    out = net(batch)
    for i in range(len(scenes)):
        assert float((out["cls"][i] - ref["cls"][i]).abs().max()) <= 2e-4
        assert torch.allclose(out["reg"][i], ref["reg"][i], rtol=1e-6, atol=5e-4)


def test_wgrad_and_gn_bwd_against_autograd(mma):
    """The two backward kernels alone against stock autograd on random data (fp32, 1e-4 relative)."""
    import lanegcn_amd  # noqa: F401
    from lanegcn_amd import _lib as L
    from lanegcn_amd import ops
    torch.manual_seed(0)
    n = 1000
    x = torch.randn(n, 128, device="cuda")
    dy = torch.randn(n, 128, device="cuda")
    g = torch.randn(128, device="cuda").abs() + 0.5
    b = torch.randn(128, device="cuda")
    res = torch.randn(n, 128, device="cuda")
    # reference on the CPU in fp64: the stock GPU GroupNorm backward is wrong for > 128 rows on this stack
    # (tools/check_aten_gn.py), which is exactly why the product has its own
    xr = x.cpu().double().requires_grad_(True)
    gr, br, rr = (t.cpu().double().requires_grad_(True) for t in (g, b, res))
    out = torch.relu(torch.nn.functional.group_norm(xr, 1, gr, br, 1e-5) + rr)
    out.backward(dy.cpu().double())
    mine = ops.gn_fwd(x, (g, b), res, True)
    assert float((mine.cpu() - out.detach().float()).abs().max()) <= 1e-5
    dx, gg, dgam, dbet = ops.gn_bwd(dy, x, mine, g, want_g=True)
    assert rel_err(dx.cpu().numpy(), xr.grad.cpu().numpy()) <= 1e-4
    assert rel_err(gg.cpu().numpy(), rr.grad.cpu().numpy()) <= 1e-6
    assert rel_err(dgam.cpu().numpy(), gr.grad.cpu().numpy()) <= 1e-4
    assert rel_err(dbet.cpu().numpy(), br.grad.cpu().numpy()) <= 1e-4
    # wgrad, IDENT relation: dW = dT^T X
    dW = ops.wgrad(n, [ops.RelSpec(x, None, L.REL_IDENT)], dy)[0]
    want = dy.double().t() @ x.double()
    assert rel_err(dW.cpu().numpy(), want.float().cpu().numpy()) <= 1e-5


def _torch_lane_conv(x, us, vs, W_ctr, W_rel, g1, b1, W2, g2, b2):
    """Stock-autograd statement of one LaneConv layer (lanegcn.py:331-362)."""
    F = torch.nn.functional
    t = F.linear(x, W_ctr)
    for u, v, w in zip(us, vs, W_rel):
        t = t.index_add(0, u, F.linear(x[v], w))
    y = torch.relu(F.group_norm(t, 1, g1, b1, 1e-5))
    z = F.group_norm(F.linear(y, W2), 1, g2, b2, 1e-5)
    return torch.relu(z + x)


@pytest.mark.parametrize("mode", ["f32", "bf16x3", "f16x2"])
@pytest.mark.parametrize("n", [70, 486])
def test_lane_conv_fn_gradients_vs_stock_autograd(mode, n):
    """LaneConvFn (fused HIP forward, composed HIP backward) against stock autograd in fp64 on a random
    multigraph: output, d input and every parameter gradient within 1e-4 of the tensor's scale."""
    import lanegcn_amd  # noqa: F401
    from lanegcn_amd import _lib as L
    from lanegcn_amd import autograd as A
    from lanegcn_amd import ops
    prev = ops.get_mma()
    ops.set_mma(mode)
    try:
        g = torch.Generator().manual_seed(n)
        rnd = lambda *s: torch.randn(*s, generator=g)
        us = [torch.randint(0, n, (m,), generator=g) for m in (3 * n, n, 0, n // 2)]
        vs = [torch.randint(0, n, (len(u),), generator=g) for u in us]
        x = rnd(n, 128)
        Ws = [rnd(128, 128) * 0.08 for _ in range(6)]       # ctr, 4 relations, ctr2
        gs = [torch.rand(128, generator=g) + 0.5 for _ in range(2)]
        bs = [rnd(128) * 0.1 for _ in range(2)]
        d_out = rnd(n, 128)
        # stock autograd in fp64 on the CPU
        P = [t.double().requires_grad_(True) for t in [x] + Ws + gs + bs]
        xr, wr, gr, br = P[0], P[1:7], P[7:9], P[9:11]
        ref = _torch_lane_conv(xr, us, vs, wr[0], wr[1:5], gr[0], br[0], wr[5], gr[1], br[1])
        ref.backward(d_out.double())
        # HIP
        D = [t.cuda().requires_grad_(True) for t in [x] + Ws + gs + bs]
        xd, wd, gd, bd = D[0], D[1:7], D[7:9], D[9:11]
        ud, vd = [u.cuda() for u in us], [v.cuda() for v in vs]
        plan, plan_t = ops.csr_build(ud, vd, n), ops.csr_build(vd, ud, n)
        rels, weights = [A.Rel(0, 0, L.REL_IDENT)], [wd[0]]
        for r in range(4):
            if plan.n_edges[r] > 0:
                rels.append(A.Rel(0, len(weights), L.REL_CSR, r))
                weights.append(wd[1 + r])
        spec = A.BlockSpec(n_rows=n, rels=rels, gn=True, relu=True, has_res=True, plan=plan, plan_t=plan_t)
        out = A.LaneConvFn.apply(spec, xd, gd[0], bd[0], wd[5], gd[1], bd[1], *weights)
        out.backward(d_out.cuda())
        assert rel_err(out.detach().cpu().numpy(), ref.detach().float().numpy()) <= 1e-4
        names = ["x", "W_ctr", "W_r0", "W_r1", "W_r2(empty)", "W_r3", "W_ctr2", "g1", "g2", "b1", "b2"]
        for name, a, b in zip(names, D, P):
            if name == "W_r2(empty)":
                assert a.grad is None or float(a.grad.abs().max()) == 0.0
                continue
            assert rel_err(a.grad.cpu().numpy(), b.grad.float().numpy()) <= 1e-4, (mode, name)
    finally:
        ops.set_mma(prev)


@pytest.mark.parametrize("mode", ["f32", "bf16x3", "f16x2"])
@pytest.mark.parametrize("n", [42, 486, 1000])
def test_row_block_gradients_vs_stock_autograd(mode, n):
    """linear_gn (IDENT relation, GN, ReLU, residual) and a weight-slice block against fp64 autograd."""
    import lanegcn_amd  # noqa: F401
    from lanegcn_amd import autograd as A
    from lanegcn_amd import ops
    F = torch.nn.functional
    prev = ops.get_mma()
    ops.set_mma(mode)
    try:
        g = torch.Generator().manual_seed(n + 1)
        x, res, d_out = (torch.randn(n, 128, generator=g) for _ in range(3))
        W = torch.randn(128, 132, generator=g) * 0.1
        gam, bet = torch.rand(128, generator=g) + 0.5, torch.randn(128, generator=g) * 0.1
        P = [t.double().requires_grad_(True) for t in (x, W, gam, bet, res)]
        ref = torch.relu(F.group_norm(F.linear(P[0], P[1][:, :128]), 1, P[2], P[3], 1e-5) + P[4])
        ref.backward(d_out.double())
        D = [t.cuda().requires_grad_(True) for t in (x, W, gam, bet, res)]
        gn = torch.nn.GroupNorm(1, 128).cuda()
        gn.weight, gn.bias = torch.nn.Parameter(D[2].detach().clone()), torch.nn.Parameter(D[3].detach().clone())
        out = A.linear_gn(D[0], D[1], gn=gn, relu=True, res=D[4], col0=0)
        out.backward(d_out.cuda())
        assert rel_err(out.detach().cpu().numpy(), ref.detach().float().numpy()) <= 1e-4
        got = [D[0].grad, D[1].grad, gn.weight.grad, gn.bias.grad, D[4].grad]
        for name, a, b in zip(["x", "W", "gamma", "beta", "res"], got, P):
            assert rel_err(a.cpu().numpy(), b.grad.float().numpy()) <= 1e-4, (mode, n, name)
    finally:
        ops.set_mma(prev)


def test_train_driver_smoke_and_resume(tmp_path):
    """train_dp.py (reference train.py semantics): a few iterations on synthetic scenes, checkpoint written in the
    reference's format, --resume restores epoch + optimizer state, --weight loads by name+shape."""
    import importlib
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    train_dp = importlib.import_module("train_dp")
    save = str(tmp_path / "run")
    rc = train_dp.main(["--max-iters", "3", "--batch-size", "2", "--save-dir", save, "--dataset-len", "8"])
    assert rc == 0
    ckpts = sorted(os.listdir(save))
    assert len(ckpts) == 1 and ckpts[0].endswith(".ckpt")
    ck = torch.load(os.path.join(save, ckpts[0]), map_location="cpu", weights_only=True)
    assert set(ck) == {"epoch", "state_dict", "opt_state"} and len(ck["state_dict"]) == 405
    assert all(not v.is_cuda for v in ck["state_dict"].values())
    rc = train_dp.main(["--max-iters", "1", "--batch-size", "2", "--save-dir", save, "--dataset-len", "8",
                        "--resume", os.path.join(save, ckpts[0])])
    assert rc == 0
    assert train_dp.main(["--eval", "--weight", os.path.join(save, ckpts[0]), "--dataset-len", "4"]) == 0


@pytest.mark.parametrize("mode", ["f32", "bf16x3", "f16x2", "bf16"])
def test_refresh_packed_matches_single_packs(mode):
    """After an in-place weight update, ONE lgcn_pack_weight_batch launch must leave every cached image (plain,
    column-offset and transposed) bit-identical to a fresh single-weight pack."""
    import lanegcn_amd  # noqa: F401
    from lanegcn_amd import ops
    prev = ops.get_mma()
    ops.set_mma(mode)
    try:
        torch.manual_seed(3)
        ws = [torch.nn.Parameter(torch.randn(128, 128, device="cuda")), torch.nn.Parameter(torch.randn(128, 384, device="cuda"))]
        imgs = [ops.packed(ws[0]), ops.packed_t(ws[0]), ops.packed(ws[1], 128, 128), ops.packed_t(ws[1], 256)]
        ptrs = [t.data_ptr() for t in imgs]
        with torch.no_grad():
            for w in ws:
                w.add_(torch.randn_like(w))          # what optimizer.step() does: in place, bumps _version
        n = ops.refresh_packed()
        assert n >= 4
        again = [ops.packed(ws[0]), ops.packed_t(ws[0]), ops.packed(ws[1], 128, 128), ops.packed_t(ws[1], 256)]
        assert [t.data_ptr() for t in again] == ptrs            # cache hits: refreshed in place, not rebuilt
        got = [t.clone() for t in again]
        for w in ws:
            ops.invalidate_packed(w)
        fresh = [ops.packed(ws[0]), ops.packed_t(ws[0]), ops.packed(ws[1], 128, 128), ops.packed_t(ws[1], 256)]
        for a, b in zip(got, fresh):
            assert torch.equal(a.view(torch.int32), b.view(torch.int32))
        assert ops.refresh_packed() == 0                        # nothing stale
    finally:
        ops.set_mma(prev)


@pytest.mark.parametrize("shape", [(1600, 32, 20), (37, 64, 10), (130, 128, 5), (3, 7, 3)])
def test_gn_cl_function_gradients_vs_fp64(shape):
    """GNCLFn (ActorNet's norm + residual + ReLU, HIP forward and backward) against fp64 CPU autograd:
    dx, dres, dgamma, dbeta."""
    import lanegcn_amd  # noqa: F401
    from lanegcn_amd import autograd as A
    torch.manual_seed(shape[0])
    n, C_, L_ = shape
    gn = torch.nn.GroupNorm(1, C_)
    with torch.no_grad():
        gn.weight.uniform_(0.5, 1.5)
        gn.bias.uniform_(-0.5, 0.5)
    x0, r0, w0 = torch.randn(n, C_, L_) * 2 + 0.3, torch.randn(n, C_, L_), torch.randn(n, C_, L_)
    for use_res in (False, True):
        for relu in (False, True):
            xd, rd = x0.double().requires_grad_(True), r0.double().requires_grad_(True)
            gd = torch.nn.GroupNorm(1, C_).double()
            gd.load_state_dict({k: v.double() for k, v in gn.state_dict().items()})
            ref = torch.nn.functional.group_norm(xd, 1, gd.weight, gd.bias, gn.eps)
            ref = ref + rd if use_res else ref
            ref = ref.relu() if relu else ref
            (ref * w0.double()).sum().backward()
            xg, rg = x0.cuda().requires_grad_(True), r0.cuda().requires_grad_(True)
            gg = torch.nn.GroupNorm(1, C_).cuda()
            gg.load_state_dict(gn.state_dict())
            out = A.gn_cl_act(xg, gg, relu=relu, res=rg if use_res else None)
            (out * w0.cuda()).sum().backward()
            assert float((out.detach().cpu().double() - ref.detach()).abs().max()) <= 2e-5
            scale = lambda t: float(t.abs().max()) + 1e-9
            assert float((xg.grad.cpu().double() - xd.grad).abs().max()) <= 2e-5 * scale(xd.grad) + 1e-6
            if use_res:
                assert float((rg.grad.cpu().double() - rd.grad).abs().max()) <= 1e-6
            assert float((gg.weight.grad.cpu().double() - gd.weight.grad).abs().max()) <= 2e-5 * scale(gd.weight.grad)
            assert float((gg.bias.grad.cpu().double() - gd.bias.grad).abs().max()) <= 2e-5 * scale(gd.bias.grad)


def test_pred_loss_hip_equals_stock_composition():
    """PredLoss on lgcn_pred_loss_fwd / _bwd (reference lanegcn.py:740-807) against the ATen composition of the same
    class: counts equal, sums to 1e-6 relative, gradients to 1e-6 -- with actors without any observed step, actors
    whose only observed step is t = 0 (dropped, :769-771), ties broken by mode order, and modes inside / outside the
    margin and ignore bands."""
    import lanegcn_amd  # noqa: F401
    from lanegcn_amd import lanegcn as M
    g = torch.Generator().manual_seed(3)
    for A in (1, 7, 333, 1600):
        gt = torch.cumsum(torch.randn(A, 30, 2, generator=g), 1).cuda()
        reg = (gt.unsqueeze(1).cpu() + 0.6 * torch.randn(A, 6, 30, 2, generator=g) * torch.rand(A, 6, 1, 1, generator=g) * 3).cuda()
        reg[:, 2] = reg[:, 1]                                        # equal distances: first minimum wins
        cls = torch.randn(A, 6, generator=g).cuda() * 0.3
        has = (torch.rand(A, 30, generator=g) > 0.3).cuda()
        has[::5] = False                                             # never observed
        if A > 3:
            has[3] = False
            has[3, 0] = True                                         # only t = 0: dropped
        res = {}
        for impl in ("stock", "hip"):
            M.PredLoss.impl = impl
            c, r = cls.clone().requires_grad_(True), reg.clone().requires_grad_(True)
            sizes = [A // 2, A - A // 2] if A > 1 else [A]
            out = {"cls": list(torch.split(c, sizes)), "reg": list(torch.split(r, sizes))}
            lo = M.Loss(M.config)(out, {"gt_preds": list(torch.split(gt, sizes)), "has_preds": list(torch.split(has, sizes))})
            lo["loss"].backward()
            res[impl] = (lo, c.grad.clone(), r.grad.clone())
        M.PredLoss.impl = "hip"
        a, b = res["stock"], res["hip"]
        assert a[0]["num_cls"] == b[0]["num_cls"] and a[0]["num_reg"] == b[0]["num_reg"], A
        for k in ("cls_loss", "reg_loss", "loss"):
            assert float(b[0][k]) == pytest.approx(float(a[0][k]), rel=2e-6, abs=1e-6), (A, k)
        assert float((a[1] - b[1]).abs().max()) <= 1e-6 * max(1.0, float(a[1].abs().max())), A
        assert float((a[2] - b[2]).abs().max()) <= 1e-6 * max(1.0, float(a[2].abs().max())), A
