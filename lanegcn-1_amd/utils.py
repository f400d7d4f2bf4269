"""Boundary helpers with the reference's names (utils.py:74-177): gpu, to_long, StepLR, Optimizer."""
import torch
from torch import optim


def gpu(data):
    """Recursively move every tensor leaf to the current CUDA device (reference utils.py:74-85)."""
    if isinstance(data, (list, tuple)):
        return [gpu(x) for x in data]
    if isinstance(data, dict):
        return {k: gpu(v) for k, v in data.items()}
    if isinstance(data, torch.Tensor):
        return data.contiguous().cuda(non_blocking=True)
    return data


def to_long(data):
    """int16 leaves -> int64 (reference utils.py:88-96); dicts are updated in place like the reference."""
    if isinstance(data, dict):
        for k in data.keys():
            data[k] = to_long(data[k])
        return data
    if isinstance(data, (list, tuple)):
        return [to_long(x) for x in data]
    if torch.is_tensor(data) and data.dtype == torch.int16:
        return data.long()
    return data


class StepLR:
    """Piecewise-constant learning rate over fractional epochs (reference utils.py:165-177)."""

    def __init__(self, lr, lr_epochs):
        assert len(lr) - len(lr_epochs) == 1
        self.lr, self.lr_epochs = lr, lr_epochs

    def __call__(self, epoch):
        return self.lr[sum(1 for e in self.lr_epochs if epoch >= e)]


class Optimizer(object):
    """torch.optim wrapper whose step(epoch) sets lr = lr_func(epoch) * coef first (reference utils.py:98-162)."""

    def __init__(self, params, config, coef=None):
        if not isinstance(params, (list, tuple)):
            params = [params]
        if coef is None:
            coef = [1.0] * len(params)
        elif isinstance(coef, (list, tuple)):
            assert len(coef) == len(params)
        else:
            coef = [coef] * len(params)
        self.coef = coef
        groups = [{"params": p, "lr": 0} for p in params]
        kind = config["opt"]
        assert kind in ("sgd", "adam", "adamw")
        if kind == "sgd":
            self.opt = optim.SGD(groups, momentum=config["momentum"], weight_decay=config["wd"])
        elif kind == "adam":
            self.opt = optim.Adam(groups, weight_decay=0)
        else:
            self.opt = optim.AdamW(groups, weight_decay=config.get("weight_decay", 0.01))
        self.lr_func = config["lr_func"]
        self.clip_grads = bool(config.get("clip_grads", False))
        if self.clip_grads:
            self.clip_low, self.clip_high = config["clip_low"], config["clip_high"]

    def zero_grad(self):
        self.opt.zero_grad()

    def step(self, epoch):
        if self.clip_grads:
            self.clip()
        lr = self.lr_func(epoch)
        for c, g in zip(self.coef, self.opt.param_groups):
            g["lr"] = lr * c
        self.opt.step()
        from . import ops
        ops.refresh_packed()     # the MFMA images of the weights, rebuilt in one launch (else: one per weight on first use)
        return lr

    def clip(self):
        for g in self.opt.param_groups:
            for p in g["params"]:
                if p.grad is not None:
                    p.grad.data.clamp_(self.clip_low, self.clip_high)

    def load_state_dict(self, opt_state):
        self.opt.load_state_dict(opt_state)

    def state_dict(self):
        return self.opt.state_dict()


def load_pretrain(net, pretrain_dict):
    """Copy every tensor whose name AND shape match into `net` (reference utils.py:51-59): this is what lets
    the published 36.000.ckpt load into this build's modules."""
    state = net.state_dict()
    for key, value in pretrain_dict.items():
        if key in state and value.size() == state[key].size():
            state[key] = value if isinstance(value, torch.Tensor) else value.data
    net.load_state_dict(state)
