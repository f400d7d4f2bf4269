"""Which launch of the folded A2M block differs between the shipped library and the compare + select ReLU build?
Run twice with the same arguments but for the library: first `save`, then `compare` (same seeds -> same inputs):
  python tools/relu_variant_diff.py default f16x2 save /tmp/x.pt ; python tools/relu_variant_diff.py relucnd f16x2 compare /tmp/x.pt"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lanegcn_amd  # noqa: E402,F401
from lanegcn_amd import _lib as L  # noqa: E402

suffix, mma, what, path = sys.argv[1:5]
if suffix != "default":
    L.LIB_PATH = os.path.join(os.path.dirname(L.LIB_PATH), "liblgcn_%s.so" % suffix)
from lanegcn_amd import data as gen  # noqa: E402
from lanegcn_amd import lanegcn as M  # noqa: E402
from lanegcn_amd import ops  # noqa: E402
from lanegcn_amd.engine import collate_flat  # noqa: E402

ops.set_mma(mma)
ops.set_guard("off")
torch.manual_seed(0)
a2m = M.A2M(M.config).cuda().eval()
scenes = gen.synth_batch("S2", seed=1)
fb = collate_flat(scenes)
g = torch.Generator().manual_seed(1)
feat = torch.randn(fb.n_nodes, 128, generator=g).relu().cuda()
actors = torch.randn(fb.n_actors, 128, generator=g).relu().cuda()
ps = ops.pairs_build(fb.node_ctrs, fb.node_off, fb.actor_ctrs, fb.actor_off, M.config["actor2map_dist"], fb.cap_a2m, True)
P = ps.count()
att0, att1 = a2m.att
out = {}
with torch.no_grad():
    # old sequence
    a = a2m.fuse_meta(feat, fb.turn, fb.control, fb.intersect)
    out["old/meta"] = a
    U, V = ops.agg_mlp_pair(att0.u_kw(a), att0.v_kw(actors))
    out["old/U0"], out["old/V0"] = U, V
    c0 = att0.ctx[0]
    m = ops.att_pairs(ps, att0.dist[0].weight, att0.dist[0].bias, ops.packed(att0.dist[2].linear.weight), M._gn(att0.dist[2].norm),
                      ops.packed(c0.linear.weight, 0, 128), U, V, M._gn(c0.norm), eps=c0.norm.eps, seg=0)
    out["old/m0"] = m[:P]
    out["old/att0"] = att0.pairs_tail(a, fb.n_actors, ps, U, V)
    # folded sequence, launch by launch
    res = ops.agg_mlp_multi([dict(a2m.meta_kw(feat, fb.turn, fb.control, fb.intersect), chain_u=att0.chain_u()),
                             att0.v_kw(actors), att1.v_kw(actors)])
    (a_f, U_f), V0_f, V1_f = res
    out["new/meta"], out["new/U0"], out["new/V0"], out["new/V1"] = a_f, U_f, V0_f, V1_f
    r = att0.pairs_tail(a_f, fb.n_actors, ps, U_f, V0_f, chain_u=att1.chain_u())
    out["new/att0"], out["new/U1"] = r
    out["new/att1"] = att1.pairs_tail(r[0], fb.n_actors, ps, r[1], V1_f)
    torch.cuda.synchronize()
out = {k: v.cpu() for k, v in out.items()}
if what == "save":
    torch.save(out, path)
    print("saved", list(out))
else:
    ref = torch.load(path)
    print("library %s vs saved (%s):" % (os.path.basename(L.LIB_PATH), mma))
    for k in out:
        d = (out[k] - ref[k]).abs()
        rows = (d > 1e-5).any(1).nonzero().flatten().numpy()
        print("  %-10s max |d| %.3g, rows with |d| > 1e-5: %d %s" % (k, float(d.max()), len(rows), rows[:10]))
    for a_, b_ in (("old/meta", "new/meta"), ("old/U0", "new/U0"), ("old/V0", "new/V0"), ("old/att0", "new/att0")):
        print("  this library, %s vs %s: equal %s" % (a_, b_, torch.equal(out[a_], out[b_])))
if what == "compare":
    for k in ("old/m0", "new/U0", "new/U1"):
        d = (out[k] - ref[k]).abs()
        rows = (d > 1e-5).any(1).nonzero().flatten().numpy()
        for r in rows[:6]:
            cols = (d[r] > 1e-5).nonzero().flatten().numpy()
            g_, w_ = out[k][r].numpy(), ref[k][r].numpy()
            nz = [c for c in cols if abs(w_[c]) > 1e-3][:6]
            print("  %s row %d: %d wrong channels (first %s); got/want at %s: %s; sum got %.5f want %.5f" %
                  (k, r, len(cols), cols[:12], nz, ["%.4f" % (g_[c] / w_[c]) for c in nz], g_.sum(), w_.sum()))

if what == "compare":
    # Which context row explains a wrong pair row?  m_p is recomputed in torch fp32 with V[c] for EVERY context row c;
    # the best match says which w the kernel used for the V gather (the pair's own w, 0, another pair's ...).
    import torch.nn.functional as F
    k = "old/m0"
    d = (out[k] - ref[k]).abs()
    rows = (d > 1e-5).any(1).nonzero().flatten()
    if len(rows):
        hi, wi = ps.hi[:P].long().cpu(), ps.wi[:P].long().cpu()
        ac, cc = ps.agt_ctrs.cpu(), ps.ctx_ctrs.cpu()
        Uc, Vc = out["old/U0"], out["old/V0"]
        cpu = lambda t: t.detach().float().cpu()
        d0, d2, c0m = att0.dist[0], att0.dist[2], att0.ctx[0]
        for r in rows[:12].tolist():
            dd = ac[hi[r]] - cc[wi[r]]
            e = F.relu(F.linear(dd, cpu(d0.weight), cpu(d0.bias)))
            e = F.relu(F.group_norm(F.linear(e, cpu(d2.linear.weight))[None], 1, cpu(d2.norm.weight), cpu(d2.norm.bias), d2.norm.eps))[0]
            t = F.linear(e, cpu(c0m.linear.weight)[:, :128]) + Uc[hi[r]]
            cand = F.relu(F.group_norm(t[None] + Vc, 1, cpu(c0m.norm.weight), cpu(c0m.norm.bias), c0m.norm.eps))      # [S, 128]
            err = (cand - out[k][r][None]).abs().max(1).values
            best = int(err.argmin())
            same_w = (wi == best).nonzero().flatten()
            print("  row %d (tile %d, row in tile %d, lane %d): own w = %d (err %.2g), best w = %d (err %.2g); pairs with that w: %s; "
                  "w of rows -64 / -8 / -2 / -1 / +1 / +64: %s" % (r, r // 64, r % 64, (r % 8) * 8, int(wi[r]), float(err[wi[r]]), best, float(err[best]),
                  same_w[:6].tolist(), [int(wi[r + o]) if 0 <= r + o < P else None for o in (-64, -8, -2, -1, 1, 64)]))
