"""ActorNet (stock Conv1d FPN, reference lanegcn.py:213-263) forward / forward+backward time on the S2 actor batch
[1600, 3, 20], to see what MIOpen costs in a training step.  Env knobs are MIOpen's own (MIOPEN_DEBUG_CONV_GEMM=0 ...);
argv[1] = "bench" turns on torch.backends.cudnn.benchmark (MIOpen find)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lanegcn_amd  # noqa: E402,F401
from lanegcn_amd import lanegcn as M  # noqa: E402


def main():
    torch.backends.cudnn.benchmark = len(sys.argv) > 1 and sys.argv[1] == "bench"
    torch.manual_seed(0)
    net = M.ActorNet(M.config).cuda().train()
    x = torch.randn(1600, 3, 20, device="cuda")

    def fwd():
        with torch.no_grad():
            return net(x)

    def fwd_bwd():
        net.zero_grad(set_to_none=True)
        net(x).square().mean().backward()

    for name, fn in (("forward (no_grad)", fwd), ("forward+backward", fwd_bwd)):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        print("%s: %.2f ms" % (name, (time.perf_counter() - t0) / 20 * 1e3), flush=True)


if __name__ == "__main__":
    main()
