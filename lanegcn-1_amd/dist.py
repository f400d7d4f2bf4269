"""One-process-per-GPU helpers over torch.distributed (backend "nccl" = RCCL over xGMI on ROCm;
"gloo" for the CPU tests).  Forward throughput shards scenes across ranks with NO data-path
collective (scenes are independent graphs); the only exchange of a training step is
the gradient average, done on ONE flat bucket (the reference averages 405 tensors through Horovod,
train.py:66-69)."""
import os
from typing import Iterable, List

import torch
import torch.distributed as dist


def env_ranks():
    """(rank, world_size, local_rank) from the torchrun environment (1-process defaults)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def is_on() -> bool:
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def barrier():
    if is_on():
        dist.barrier()


def max_over_ranks(value: float, device="cpu") -> float:
    """Slowest rank's time: the whole job is as fast as its slowest shard."""
    if not is_on():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float, device="cpu") -> float:
    if not is_on():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def job_throughput(units_this_rank: float, elapsed_this_rank: float, device="cpu") -> float:
    """Whole-job units/s: all ranks' units over the MAX elapsed time."""
    return sum_over_ranks(units_this_rank, device) / max_over_ranks(elapsed_this_rank, device)


def shard(n_items: int, rank: int, world: int, seed: int = 0, epoch: int = 0, shuffle: bool = True,
          drop_last: bool = True) -> List[int]:
    """Indices of this rank's shard with DistributedSampler semantics (train.py:119-131): one
    seeded permutation shared by all ranks, rank r takes items r, r+world, ..."""
    g = torch.Generator()
    g.manual_seed(seed + epoch)
    idx = torch.randperm(n_items, generator=g).tolist() if shuffle else list(range(n_items))
    if drop_last:
        idx = idx[: (n_items // world) * world]
    else:
        pad = (-len(idx)) % world
        idx = idx + idx[:pad]
    return idx[rank::world]


def broadcast_parameters(tensors: Iterable[torch.Tensor], src: int = 0):
    """hvd.broadcast_parameters(net.state_dict(), 0) (train.py:96,145) as one flat broadcast."""
    tensors = [t for t in tensors]
    if not is_on() or not tensors:
        return
    flat = torch.cat([t.detach().reshape(-1).float() for t in tensors])
    dist.broadcast(flat, src)
    off = 0
    with torch.no_grad():
        for t in tensors:
            n = t.numel()
            t.copy_(flat[off:off + n].view_as(t))
            off += n


def allreduce_mean_grads(params: Iterable[torch.nn.Parameter]):
    """Average gradients over ranks (Horovod DistributedOptimizer semantics, train.py:66-69) with one
    all-reduce over a single flat fp32 bucket (14.8 MB for the full net).

    The bucket covers EVERY parameter that requires grad, in parameter order, zeros standing in for a gradient
    this rank does not have: which parameters receive a gradient depends on the rank's batch (a relation without
    edges skips its weights, lanegcn.py:343-354; an Att block without context skips dist / query / ctx, :664-670),
    and ranks that disagreed on the bucket's length would hang or corrupt the collective.  Afterwards every such
    parameter holds the average (Horovod averages zeros in the same way)."""
    ps = [p for p in params if p.requires_grad]
    if not is_on() or not ps:
        return
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).float() for p in ps])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat.div_(dist.get_world_size())
    off = 0
    for p in ps:
        n = p.numel()
        g = flat[off:off + n].view_as(p)
        if p.grad is None:
            p.grad = g.to(p.dtype).clone()
        else:
            p.grad.copy_(g)
        off += n


class GradBucket:
    """Persistent flat fp32 gradient bucket: every parameter's `.grad` is a VIEW of one buffer, so backward accumulates
    straight into it, `zero()` is one fill, and the average over ranks is ONE in-place all-reduce -- no `torch.cat` of
    405 gradients and no 405 copies back per step (what allreduce_mean_grads does without a bucket).
    Same semantics as allreduce_mean_grads: the bucket covers every parameter that requires grad, in parameter order,
    and a parameter without a gradient on this rank contributes zeros.  Use `zero()` instead of
    `optimizer.zero_grad()` (set_to_none would drop the views; `attach()` re-installs them if something did)."""

    def __init__(self, params: Iterable[torch.nn.Parameter]):
        self.params = [p for p in params if p.requires_grad]
        if any(p.dtype != torch.float32 for p in self.params):
            raise TypeError("GradBucket holds fp32 gradients")
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device if self.params else torch.device("cpu")
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        self.views, off = [], 0
        for p in self.params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        self.attach()

    def attach(self):
        """(Re-)install the views as the parameters' gradients; a gradient found elsewhere is copied in first."""
        for p, v in zip(self.params, self.views):
            if p.grad is not v:
                if p.grad is not None:
                    v.copy_(p.grad)
                p.grad = v

    def zero(self):
        self.attach()
        self.flat.zero_()

    def allreduce_mean(self):
        self.attach()
        if is_on():
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
            self.flat.div_(dist.get_world_size())


def gather_metrics(metrics: dict) -> dict:
    """The reference's `sync` (train.py:245-259, an MPI allgather + merge): scalars are summed over ranks, lists of
    per-batch arrays are concatenated in rank order.  Single process: returned as is."""
    if not (torch.distributed.is_available() and torch.distributed.is_initialized()) or torch.distributed.get_world_size() == 1:
        return metrics
    parts = [None] * torch.distributed.get_world_size()
    torch.distributed.all_gather_object(parts, metrics)
    out = {}
    for part in parts:
        for k, v in (part or {}).items():
            if isinstance(v, list):
                out.setdefault(k, []).extend(v)
            else:
                out[k] = out.get(k, 0.0) + v
    return out
