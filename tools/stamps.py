"""Phase timeline of the LaneConv launch from the diagnostic library (make -C lanegcn-1_amd/csrc stamps).
s_memtime ticks are shader cycles; prints the median over workgroups of each phase, per role."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lanegcn_amd  # noqa: E402,F401
from lanegcn_amd import _lib as L  # noqa: E402

L.LIB_PATH = os.path.join(os.path.dirname(L.LIB_PATH), "liblgcn_stamps.so")
from lanegcn_amd import data as gen  # noqa: E402
from lanegcn_amd import lanegcn as M  # noqa: E402
from lanegcn_amd import ops  # noqa: E402
from lanegcn_amd.engine import collate_flat  # noqa: E402


def main():
    rb = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    ops.set_mma(sys.argv[2] if len(sys.argv) > 2 else "bf16x3")
    torch.manual_seed(0)
    net = M.MapNet(M.config).cuda().eval()
    fb = collate_flat(gen.synth_batch("S2", seed=100))
    with torch.no_grad():
        g64, _ = ops.graph_gather_indices(fb.idx_local, fb.seg_off, fb.seg_base)
        plan = ops.csr_build([g64[a:b] for (a, b), _ in fb.rel_slices], [g64[a:b] for _, (a, b) in fb.rel_slices], fb.n_nodes)
        N = fb.n_nodes
        x = torch.randn(N, 128, device="cuda")
        fuse, keys = net.fuse, M.rel_keys(6)
        rels = [ops.RelSpec(x, ops.packed(fuse["ctr"][0].weight), L.REL_IDENT)]
        rels += [ops.RelSpec(x, ops.packed(fuse[k][0].weight), L.REL_CSR, r) for r, k in enumerate(keys)]
        c2 = fuse["ctr2"][0]
        n_tiles = (N + 16 * rb - 1) // (16 * rb)
        stamps = torch.zeros(n_tiles * 128, dtype=torch.int64, device="cuda")
        full = L.F_GN1 | L.F_RELU1 | L.F_GEMM2 | L.F_GN2 | L.F_RES | L.F_RELU2
        for _ in range(3):
            ops.agg_mlp(N, rels, full, rowptr=plan.rowptr, col=plan.col, n_rel_csr=plan.n_rel,
                        gn1=(fuse["norm"][0].weight, fuse["norm"][0].bias), wp2=ops.packed(c2.linear.weight),
                        gn2=(c2.norm.weight, c2.norm.bias), res=x, tile_rb=rb, out_pre=stamps)
        torch.cuda.synchronize()
    st = stamps.cpu().numpy().reshape(n_tiles, 2, 64).astype(np.float64)
    t0 = st[:, 0, 0:1]
    mf, ga = st[:, 0, :] - t0, st[:, 1, :] - t0
    med = lambda a: float(np.median(a))
    print("rb=%d mma=%s tiles=%d  (cycles, median over workgroups; 100 MHz memtime? see total)" % (rb, ops.get_mma(), n_tiles))
    print("kernel span per WG (stamp46 of the row wave - stamp0): %.0f" % med(ga[:, 46]))
    print("  index preload + act (0->1): %.0f" % med(mf[:, 1]))
    print("  index ready -> loop start (1->3; relation 0 is fetched before the index slice): MFMA %.0f, gather %.0f" % (
        med(mf[:, 3] - mf[:, 1]), med(ga[:, 3] - ga[:, 1])))
    work_m = [med(mf[:, 4 + 2 * i] - mf[:, 3 + 2 * i]) for i in range(15)]
    wait_m = [med(mf[:, 5 + 2 * i] - mf[:, 4 + 2 * i]) for i in range(15)]
    work_g = [med(ga[:, 4 + 2 * i] - ga[:, 3 + 2 * i]) for i in range(15)]
    wait_g = [med(ga[:, 5 + 2 * i] - ga[:, 4 + 2 * i]) for i in range(15)]
    print("  per pass MFMA work  :", " ".join("%5.0f" % v for v in work_m))
    print("  per pass MFMA wait  :", " ".join("%5.0f" % v for v in wait_m))
    print("  per pass gather work:", " ".join("%5.0f" % v for v in work_g))
    print("  per pass gather wait:", " ".join("%5.0f" % v for v in wait_g))
    print("  loop total (3->33): %.0f" % med(mf[:, 33] - mf[:, 3]))
    print("  epilogue, MFMA wave : acc store (33->40) %.0f | barrier %.0f | wp2 prefetch (41->42) %.0f | barrier %.0f | gemm2 (43->44) %.0f | barrier %.0f" % (
        med(mf[:, 40] - mf[:, 33]), med(mf[:, 41] - mf[:, 40]), med(mf[:, 42] - mf[:, 41]), med(mf[:, 43] - mf[:, 42]),
        med(mf[:, 44] - mf[:, 43]), med(mf[:, 45] - mf[:, 44])))
    print("  epilogue, row wave  : wait (33->41) %.0f | row phase 1 (41->42) %.0f | barrier+gemm2 wait (42->45) %.0f | row phase 2 (45->46) %.0f" % (
        med(ga[:, 41] - ga[:, 33]), med(ga[:, 42] - ga[:, 41]), med(ga[:, 45] - ga[:, 42]), med(ga[:, 46] - ga[:, 45])))
    start = st[:, 0, 0]
    end = st[:, 1, 46]
    print("  launch skew: first WG start -> last WG start %.0f ; first start -> last end %.0f" % (start.max() - start.min(), end.max() - start.min()))


if __name__ == "__main__":
    main()
