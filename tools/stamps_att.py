"""Phase timeline of k_att_pairs_ws from the diagnostic library (make -C lanegcn-1_amd/csrc stamps): s_memtime deltas,
medians over workgroups, for the A2A / M2A / A2M pair sets of a synthetic S2 batch.
Usage: python tools/stamps_att.py [a2a|m2a|a2m] [seg]"""
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
    which = sys.argv[1] if len(sys.argv) > 1 else "a2a"
    seg = int(sys.argv[2]) if len(sys.argv) > 2 else (0 if which == "a2m" else 16)
    ops.set_mma("f16x2")
    torch.manual_seed(0)
    att = M.Att(128, 128).cuda().eval()
    fb = collate_flat(gen.synth_batch("S2", seed=100, n_scenes=32))
    cfg = M.config
    s = {"a2m": (fb.node_ctrs, fb.node_off, fb.actor_ctrs, fb.actor_off, cfg["actor2map_dist"], fb.cap_a2m),
         "m2a": (fb.actor_ctrs, fb.actor_off, fb.node_ctrs, fb.node_off, cfg["map2actor_dist"], fb.cap_a2m),
         "a2a": (fb.actor_ctrs, fb.actor_off, fb.actor_ctrs, fb.actor_off, cfg["actor2actor_dist"], fb.cap_a2a)}[which]
    lib = L.load()
    with torch.no_grad():
        ps = ops.pairs_build(*s)
        P = ps.count()
        T, S = s[0].shape[0], s[2].shape[0]
        U = torch.randn(T, 128, device="cuda")
        V = torch.randn(S, 128, device="cuda")
        c0 = att.ctx[0]
        n_wg = 512
        stamps = torch.zeros(n_wg * 2 * 32, dtype=torch.int64, device="cuda")
        lib.lgcn_debug_att_stamps.argtypes = [C.c_void_p]
        lib.lgcn_debug_att_stamps(C.c_void_p(stamps.data_ptr()))
        args = (ps, att.dist[0].weight, att.dist[0].bias, ops.packed(att.dist[2].linear.weight), M._gn(att.dist[2].norm),
                ops.packed(c0.linear.weight, 0, 128), U, V, M._gn(c0.norm))
        for _ in range(3):
            stamps.zero_()
            ops.att_pairs(*args, seg=seg)
        torch.cuda.synchronize()
    st = stamps.cpu().numpy().reshape(n_wg, 2, 32).astype(np.float64)
    print("%s: P=%d tiles=%d seg=%d" % (which, P, (P + 63) // 64, seg))
    names = ["start->requests", "barrier", "e0 planes (+barrier)", "GEMM1", "tile store + barrier", "row phase", "barrier",
             "U+V", "GEMM2", "tile store + barrier", "row phase", "pieces", "barrier"]
    for role, nm in ((0, "wave 0"), (1, "wave 7")):
        s_ = st[:, role, :]
        live = s_[:, 3] > 0
        d = np.diff(s_[live], axis=1)
        print(" %s (%d workgroups with a tile):" % (nm, int(live.sum())),
              " | ".join("%s %.0f" % (n, float(np.median(d[:, i]))) for i, n in enumerate(names)))
        two = s_[:, 16] > 0
        if two.any():
            d2 = np.diff(s_[two], axis=1)
            print("   second tile (%d workgroups):" % int(two.sum()), " ".join("%.0f" % float(np.median(d2[:, i])) for i in range(13, 24)))
        last = (s_[live] > 0).sum(1) - 1
        print("   whole workgroup %.0f cycles" % float(np.median(s_[live][np.arange(int(live.sum())), last] - s_[live][:, 0])))


if __name__ == "__main__":
    main()
