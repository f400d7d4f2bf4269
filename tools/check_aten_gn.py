"""Stock ATen GroupNorm(1, C) backward for 2-D [rows, C] inputs: GPU vs CPU (affine grads)."""
import torch
import torch.nn.functional as F
torch.manual_seed(0)
for rows in (8, 42, 64, 65, 128, 252, 1000, 5000):
    x = torch.randn(rows, 128); w = torch.rand(128) + 0.5; b = torch.randn(128); dy = torch.randn(rows, 128)
    res = {}
    for dev in ("cpu", "cuda"):
        xx, ww, bb = (t.to(dev).clone().requires_grad_(True) for t in (x, w, b))
        y = F.group_norm(xx, 1, ww, bb, 1e-5)
        y.backward(dy.to(dev))
        res[dev] = (y.detach().cpu(), xx.grad.cpu(), ww.grad.cpu(), bb.grad.cpu())
    e = [float((a - c).abs().max() / (a.abs().max() + 1e-12)) for a, c in zip(res["cpu"], res["cuda"])]
    print("rows %5d  rel err  y %.1e  dx %.1e  dgamma %.1e  dbeta %.1e" % (rows, *e))
print("3-D inputs [N, C, L] (ActorNet shapes)")
for shape in ((42, 32, 20), (1600, 32, 20), (1600, 64, 10), (1600, 128, 5), (1600, 128, 20), (200, 128, 20)):
    x = torch.randn(*shape); C = shape[1]; w = torch.rand(C) + 0.5; b = torch.randn(C); dy = torch.randn(*shape)
    res = {}
    for dev in ("cpu", "cuda"):
        xx, ww, bb = (t.to(dev).clone().requires_grad_(True) for t in (x, w, b))
        y = F.group_norm(xx, 1, ww, bb, 1e-5); y.backward(dy.to(dev))
        res[dev] = (xx.grad.cpu(), ww.grad.cpu(), bb.grad.cpu())
    e = [float((a - c).abs().max() / (a.abs().max() + 1e-12)) for a, c in zip(res["cpu"], res["cuda"])]
    print("shape %-16s rel err dx %.1e dgamma %.1e dbeta %.1e" % (shape, *e))
