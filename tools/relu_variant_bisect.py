"""Which launch of the A2M block gives wrong rows on the compare + select ReLU build (tools/relu_variant_check.py)?
Every launch type of the block is repeated REPS times on the same inputs and compared with its first result bit for
bit.  Usage: python tools/relu_variant_bisect.py [lib suffix, default relucnd] [reps] [mma]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lanegcn_amd  # noqa: E402,F401
from lanegcn_amd import _lib as L  # noqa: E402

suffix = sys.argv[1] if len(sys.argv) > 1 else "relucnd"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
mma = sys.argv[3] if len(sys.argv) > 3 else "bf16x3"
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
cfg = M.config
ps = ops.pairs_build(fb.node_ctrs, fb.node_off, fb.actor_ctrs, fb.actor_off, cfg["actor2map_dist"], fb.cap_a2m, True)
att = a2m.att[0]
print("library %s, mode %s, %d nodes, %d actors, %d pairs" % (os.path.basename(L.LIB_PATH), mma, fb.n_nodes, fb.n_actors, ps.count()), flush=True)


def repeat(name, fn):
    first = [t.clone() for t in fn()]
    torch.cuda.synchronize()
    bad_runs, rows_mod = 0, set()
    worst = 0.0
    for _ in range(reps):
        out = fn()
        torch.cuda.synchronize()
        hit = False
        for a, b in zip(out, first):
            if not torch.equal(a, b):
                hit = True
                d = (a != b).any(1).nonzero().flatten().cpu().numpy()
                rows_mod |= set((d % 64).tolist())
                worst = max(worst, float((a - b).abs().max()))
        bad_runs += hit
    print("%-34s %3d / %d runs differ from the first; rows %% 64: %s; max |d| %.3g" %
          (name, bad_runs, reps, sorted(rows_mod)[:24], worst), flush=True)
    return first


with torch.no_grad():
    f1 = repeat("meta (GN1+ReLU1, rank-4)", lambda: [a2m.fuse_meta(feat, fb.turn, fb.control, fb.intersect)])[0]
    uv = repeat("U | V dual launch", lambda: list(ops.agg_mlp_pair(att.u_kw(f1), att.v_kw(actors))))
    repeat("U alone", lambda: [ops.agg_mlp(**att.u_kw(f1))])
    repeat("V alone (1600 rows)", lambda: [ops.agg_mlp(**att.v_kw(actors))])
    P = ps.count()
    c0 = att.ctx[0]
    m = repeat("pair MLP", lambda: [ops.att_pairs(ps, att.dist[0].weight, att.dist[0].bias, ops.packed(att.dist[2].linear.weight),
                                                  M._gn(att.dist[2].norm), ops.packed(c0.linear.weight, 0, 128), uv[0], uv[1],
                                                  M._gn(c0.norm), eps=c0.norm.eps, seg=0)[:P]])[0]
    lin = att.linear
    rels = [ops.RelSpec(f1, ops.packed(att.agt.weight)), ops.RelSpec(m, ops.packed(att.ctx[1].weight), L.REL_RANGE)]
    repeat("tail (2 rel, GN, GEMM2, GN, res)", lambda: [ops.agg_mlp(fb.n_nodes, rels, M._FULL, rowptr=ps.rowptr, gn1=M._gn(att.norm),
                                                                  wp2=ops.packed(lin.linear.weight), gn2=M._gn(lin.norm), res=f1,
                                                                  eps=att.norm.eps)])
    repeat("plain Linear+GN+ReLU 10368 rows", lambda: [ops.agg_mlp(fb.n_nodes, [ops.RelSpec(f1, ops.packed(att.agt.weight))],
                                                                   L.F_GN1 | L.F_RELU1, gn1=M._gn(att.norm))])
    for rb in (1, 2, 3):
        repeat("Linear+GN+ReLU, tile_rb=%d" % rb, lambda: [ops.agg_mlp(fb.n_nodes, [ops.RelSpec(f1, ops.packed(att.agt.weight))],
                                                                       L.F_GN1 | L.F_RELU1, gn1=M._gn(att.norm), tile_rb=rb)])

    # ---- the same launches back to back (no synchronisation in between), as a forward issues them: which
    # intermediate is the first to differ from the first run's?
    def chain():
        a = a2m.fuse_meta(feat, fb.turn, fb.control, fb.intersect)
        U, V = ops.agg_mlp_pair(att.u_kw(a), att.v_kw(actors))
        mm = ops.att_pairs(ps, att.dist[0].weight, att.dist[0].bias, ops.packed(att.dist[2].linear.weight), M._gn(att.dist[2].norm),
                           ops.packed(c0.linear.weight, 0, 128), U, V, M._gn(c0.norm), eps=c0.norm.eps, seg=0)
        r2 = [ops.RelSpec(a, ops.packed(att.agt.weight)), ops.RelSpec(mm, ops.packed(att.ctx[1].weight), L.REL_RANGE)]
        o = ops.agg_mlp(fb.n_nodes, r2, M._FULL, rowptr=ps.rowptr, gn1=M._gn(att.norm), wp2=ops.packed(lin.linear.weight),
                        gn2=M._gn(lin.norm), res=a, eps=att.norm.eps)
        return {"meta": a, "U": U, "V": V, "m": mm[:P], "tail": o}

    ref = {k: v.clone() for k, v in chain().items()}
    torch.cuda.synchronize()
    firsts = {}
    for r in range(reps):
        out = chain()
        torch.cuda.synchronize()
        for k in ("meta", "U", "V", "m", "tail"):
            if not torch.equal(out[k], ref[k]):
                d = (out[k] != ref[k]).any(1).nonzero().flatten().cpu().numpy()
                firsts.setdefault(k, []).append((r, len(d), sorted(set((d % 64).tolist()))[:12]))
                break
    print("back-to-back chain, %d runs; first differing intermediate per failing run:" % reps, flush=True)
    for k, v in firsts.items():
        print("  %-5s in %d runs, e.g. (run, rows, rows %% 64): %s" % (k, len(v), v[:4]), flush=True)
    if not firsts:
        print("  none", flush=True)
