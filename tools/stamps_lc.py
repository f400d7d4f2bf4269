"""Phase timeline of k_lc_tile (weight-stationary LaneConv) from the diagnostic library
(make -C lanegcn-1_amd/csrc stamps): s_memtime ticks = shader cycles; medians over workgroups.
Usage: python tools/stamps_lc.py [groups] [mma]"""
import ctypes as C
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
    groups = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    ops.set_mma(sys.argv[2] if len(sys.argv) > 2 else "f16x2")
    scenes = int(sys.argv[3]) if len(sys.argv) > 3 else 32
    variant = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    torch.manual_seed(0)
    net = M.MapNet(M.config).cuda().eval()
    fb = collate_flat(gen.synth_batch("S2", seed=100, n_scenes=scenes))
    lib = L.load()
    with torch.no_grad():
        g64, _ = ops.graph_gather_indices(fb.idx_local, fb.seg_off, fb.seg_base)
        plan = ops.csr_build([g64[a:b] for (a, b), _ in fb.rel_slices], [g64[a:b] for _, (a, b) in fb.rel_slices], fb.n_nodes)
        x = torch.randn(fb.n_nodes, 128, device="cuda").relu()
        fuse, keys = net.fuse, M.rel_keys(6)
        wps = [ops.packed(fuse["ctr"][0].weight)] + [ops.packed(fuse[k][0].weight) for k in keys]
        c2 = fuse["ctr2"][0]
        lcp = ops.lc_plan(plan, n_groups=groups, variant=variant)
        n_wg = ((fb.n_nodes + lcp.rows_per_block - 1) // lcp.rows_per_block) * (len(lcp.gstart) - 1)
        stamps = torch.zeros(n_wg * 2 * 64, dtype=torch.int64, device="cuda")
        lib.lgcn_debug_lc_stamps.argtypes = [C.c_void_p]
        lib.lgcn_debug_lc_stamps(C.c_void_p(stamps.data_ptr()))
        part = ops.lc_part(lcp)
        for _ in range(3):
            ops.laneconv_fwd(x, lcp, wps, M._gn(fuse["norm"][0]), ops.packed(c2.linear.weight), M._gn(c2.norm), part=part)
        torch.cuda.synchronize()
    raw = stamps.cpu().numpy().reshape(n_wg, 2, 64)
    rt = raw[:, 0, 63] - raw[:, 0, 62]                    # 100 MHz reference clock over the workgroup
    cy = raw[:, 0, 61] - raw[:, 0, 0]
    st = raw[:, :, :60].astype(np.float64)                # slots 61..63: end stamp + the reference clock pair
    units = [lcp.gstart[i + 1] - lcp.gstart[i] for i in range(len(lcp.gstart) - 1)]
    print("groups=%d (units per group %s) mma=%s workgroups=%d" % (groups, units, ops.get_mma(), n_wg))
    g = np.arange(n_wg) % len(units)      # not exact under the XCD remap; use the longest common prefix
    nu = min(units)
    for role, name in ((0, "wave 0"), (1, "wave 7")):
        s = st[:, role, :]
        d = np.diff(s, axis=1)
        med = lambda a: float(np.median(a))
        names = ["start->header", "header->rows stored", "barrier"] + ["unit %d" % k for k in range(nu)]
        print(" %s:" % name, " | ".join("%s %.0f" % (n, med(d[:, i])) for i, n in enumerate(names)))
    if os.environ.get("LGCN_STAMPS_RAW"):      # every delta, in stamp order (for the LGCN_EXP_LC diagnostic knobs)
        for role, name in ((0, "wave 0"), (1, "wave 7")):
            d = np.diff(st[:, role, :], axis=1)
            cnt = int(np.median((st[:, role, :] > 0).sum(1)))
            print(" %s raw:" % name, " ".join("%.0f" % float(np.median(d[:, i])) for i in range(cnt - 1)))
    # tail stamps are counted from the end of each workgroup's record
    for role, name in ((0, "wave 0"), (1, "wave 7")):
        s = st[:, role, :]
        last = (s > 0).sum(1) - 1
        tail = np.stack([s[np.arange(n_wg), last - j] for j in range(4)], 1)      # rows stored, first tiles in LDS, epilogue start, loop done
        dd = -np.diff(tail, axis=1)
        print(" %s tail: last barrier %.0f | accumulators -> tiles %.0f | rest of the epilogue %.0f ; whole workgroup %.0f" % (
            name, np.median(dd[:, 2]), np.median(dd[:, 1]), np.median(dd[:, 0]), np.median(tail[:, 0] - s[:, 0])))
    start, end = st[:, 0, 0], st[np.arange(n_wg), 0, (st[:, 0, :] > 0).sum(1) - 1]
    print(" shader clock %.2f GHz (cycles / reference time, median over workgroups)" % float(np.median(cy / np.maximum(rt, 1)) * 0.1))
    print(" first start -> last start %.0f ; first start -> last end %.0f cycles" % (start.max() - start.min(), end.max() - start.min()))


if __name__ == "__main__":
    main()
