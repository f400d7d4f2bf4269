"""Row-block launches (lgcn_agg_mlp, KIND 0: Linear / Att roles) at a forced tile height against the library's pick:
plain Linear + GN + ReLU, two stages with residual, RANGE / RANGE16 segment sums, chained U / V outputs, several
problems in one launch.  One tile height per process (a faulting kernel takes only that process down):
    python tools/check_tile_rb.py <rb> [mma]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lanegcn_amd  # noqa: E402,F401
from lanegcn_amd import _lib as L  # noqa: E402
from lanegcn_amd import ops  # noqa: E402


def run(rb, verbose=True):
    """Names of the cases in which tile height rb differs from the library's pick (empty: all equal)."""
    g = torch.Generator().manual_seed(5)
    w = lambda: ops.packed((torch.randn(128, 128, generator=g) * 0.09).cuda())
    gn = lambda: ((1 + 0.1 * torch.randn(128, generator=g)).cuda(), (0.1 * torch.randn(128, generator=g)).cuda())
    full = L.F_GN1 | L.F_RELU1 | L.F_GEMM2 | L.F_GN2 | L.F_RES | L.F_RELU2
    bad = []

    def check(name, fn):
        a, b = fn(0), fn(rb)
        torch.cuda.synchronize()
        a, b = (a if isinstance(a, tuple) else (a,)), (b if isinstance(b, tuple) else (b,))
        flat = lambda t: [x for y in t for x in (y if isinstance(y, tuple) else (y,))]
        err = max(float((x - y).abs().max()) for x, y in zip(flat(a), flat(b)))
        if verbose:
            print("  %-46s max |rb %d - pick| = %.3g" % (name, rb, err), flush=True)
        if not err <= 2e-5:
            bad.append(name)

    for n in (37, 500, 1041):
        if verbose:
            print("n_rows %d" % n, flush=True)
        x = torch.randn(n, 128, generator=g).cuda()
        w1, w2, wq, wu, wv, g1, g2, gq = w(), w(), w(), w(), w(), gn(), gn(), gn()
        check("Linear + GN + ReLU", lambda r: ops.agg_mlp(n, [ops.RelSpec(x, w1)], L.F_GN1 | L.F_RELU1, gn1=g1, tile_rb=r))
        check("two stages + residual", lambda r: ops.agg_mlp(n, [ops.RelSpec(x, w1)], full, gn1=g1, wp2=w2, gn2=g2, res=x, tile_rb=r))
        # segment sums: P pair rows, sorted target ids
        P = 6 * n + 11
        hi = torch.sort(torch.randint(0, n, (P,), generator=g)).values.to(torch.int32)
        rowptr = torch.searchsorted(hi.long(), torch.arange(n + 1)).to(torch.int32).cuda()
        m = torch.randn(P, 128, generator=g).cuda()
        split = ops.get_mma() != "f32"                 # RANGE16 (piece sums of the pair kernels) exists in the 16-bit-plane modes
        for mode, nm in ((L.REL_RANGE, "RANGE"),) + (((L.REL_RANGE16, "RANGE16"),) if split else ()):
            check("IDENT + %s relation, two stages" % nm,
                  lambda r, mode=mode: ops.agg_mlp(n, [ops.RelSpec(x, w1), ops.RelSpec(m, w2, mode)], full, gn1=g1, wp2=wq, gn2=g2,
                                                   res=x, rowptr=rowptr, tile_rb=r))
        check("IDENT + RANGE(16), two stages, chained U and V (an Att tail)",
              lambda r: ops.agg_mlp(n, [ops.RelSpec(x, w1), ops.RelSpec(m, w2, L.REL_RANGE16 if split else L.REL_RANGE)], full, gn1=g1, wp2=wq, gn2=g2, res=x,
                                    rowptr=rowptr, chain_u=(wq, gq, wu), chain_v=wv, tile_rb=r))
        x4 = (torch.randn(n, 2, generator=g).cuda(), torch.randn(n, generator=g).cuda(), torch.randn(n, generator=g).cuda())
        w4 = (torch.randn(128, 4, generator=g) * 0.1).cuda()
        check("rank-4 update + GN + ReLU + residual, chained U (A2M.meta)",
              lambda r: ops.agg_mlp(n, [ops.RelSpec(x, w1)], L.F_GN1 | L.F_RES | L.F_RELU1, gn1=g1, res=x, x4=x4, w4=w4,
                                    chain_u=(wq, gq, wu), tile_rb=r))
        check("chained U and V behind two stages",
              lambda r: ops.agg_mlp(n, [ops.RelSpec(x, w1)], full, gn1=g1, wp2=w2, gn2=g2, res=x, chain_u=(wq, gq, wu), chain_v=wv, tile_rb=r))
        check("chained U behind one stage", lambda r: ops.agg_mlp(n, [ops.RelSpec(x, w1)], L.F_GN1 | L.F_RELU1, gn1=g1, chain_u=(wq, gq, wu), tile_rb=r))
        xb = torch.randn(n // 3 + 5, 128, generator=g).cuda()
        pa = dict(n_rows=n, rels=[ops.RelSpec(x, w1)], flags=L.F_GN1 | L.F_RELU1, gn1=g1, chain_u=(wq, gq, wu))
        pb = dict(n_rows=xb.shape[0], rels=[ops.RelSpec(xb, w2)], flags=0)
        check("multi: chained + plain", lambda r: tuple(ops.agg_mlp_multi([dict(pa, tile_rb=r), dict(pb, tile_rb=r)])))
    return bad


def main():
    rb = int(sys.argv[1])
    ops.set_mma(sys.argv[2] if len(sys.argv) > 2 else "f16x2")
    bad = run(rb)
    print("tile_rb %d: %s" % (rb, "FAILED " + str(bad) if bad else "ok"))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
