"""CPU, world_size 2, gloo: the N > 1 plumbing of bench.py / DP training (rank sharding, max-over-ranks
timing, whole-job throughput, flat gradient average, parameter broadcast)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import lanegcn_amd  # noqa: F401
    from lanegcn_amd import dist as D
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        assert D.env_ranks() == (rank, world, rank)
        # timing: slowest rank wins; throughput: all units / max time
        assert D.max_over_ranks(1.0 + rank) == float(world)
        assert D.job_throughput(32 * 10, 1.0 + rank) == pytest.approx(32 * 10 * world / world)
        # shards: disjoint, equal size, same permutation on every rank
        mine = D.shard(101, rank, world, seed=5, epoch=2)
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        flat = sum(gathered, [])
        assert len(set(flat)) == len(flat) == (101 // world) * world
        assert all(len(g) == 101 // world for g in gathered)
        # broadcast: rank 0's weights everywhere
        torch.manual_seed(rank)
        net = torch.nn.Sequential(torch.nn.Linear(8, 8), torch.nn.GroupNorm(1, 8))
        D.broadcast_parameters(net.state_dict().values())
        torch.manual_seed(0)
        ref = torch.nn.Sequential(torch.nn.Linear(8, 8), torch.nn.GroupNorm(1, 8))
        for a, b in zip(net.parameters(), ref.parameters()):
            assert torch.equal(a, b)
        # gradient average == mean of the per-rank gradients (Horovod average semantics)
        x = torch.full((4, 8), float(rank + 1))
        net(x).sum().backward()
        local = [p.grad.clone() for p in net.parameters()]
        D.allreduce_mean_grads(net.parameters())
        every = [None] * world
        dist.all_gather_object(every, local)
        for i, p in enumerate(net.parameters()):
            want = sum(g[i] for g in every) / world
            assert torch.allclose(p.grad, want, atol=1e-6)
        # ranks that disagree on WHICH parameters have a gradient (a relation without edges on one rank's batch,
        # reference lanegcn.py:343-354): the bucket covers every parameter, zeros for the missing ones
        two = torch.nn.ModuleList([torch.nn.Linear(4, 4, bias=False), torch.nn.Linear(4, 4, bias=False)])
        for p_ in two.parameters():
            torch.nn.init.constant_(p_, 0.5)
        used = two[0] if rank == 0 else two[1]          # rank 0 only touches layer 0, rank 1 only layer 1
        used(torch.ones(2, 4) * (rank + 1)).sum().backward()
        assert (two[1].weight.grad is None) == (rank == 0)
        D.allreduce_mean_grads(two.parameters())
        assert torch.allclose(two[0].weight.grad, torch.full((4, 4), 2.0 * 1 / world))      # rank 0: sum over 2 rows of 1
        assert torch.allclose(two[1].weight.grad, torch.full((4, 4), 2.0 * 2 / world))      # rank 1: 2 rows of 2
        # the persistent bucket: gradients are views of one buffer, same averages, in place, step after step
        torch.manual_seed(0)
        net2 = torch.nn.Sequential(torch.nn.Linear(8, 8), torch.nn.GroupNorm(1, 8), torch.nn.Linear(8, 3))
        net2[2].weight.requires_grad_(rank >= 0)
        bucket = D.GradBucket(net2.parameters())
        assert bucket.flat.numel() == sum(p.numel() for p in net2.parameters())
        for step in range(2):
            bucket.zero()
            net2(torch.full((4, 8), float(rank + 1 + step))).sum().backward()
            assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(bucket.params, bucket.views))
            local = [p.grad.clone() for p in net2.parameters()]
            bucket.allreduce_mean()
            every = [None] * world
            dist.all_gather_object(every, local)
            for i, p in enumerate(net2.parameters()):
                assert torch.allclose(p.grad, sum(g[i] for g in every) / world, atol=1e-6)
        net2.zero_grad(set_to_none=True)                  # something dropped the views: attach() brings them back
        bucket.zero()
        assert all(p.grad is v for p, v in zip(bucket.params, bucket.views)) and float(bucket.flat.abs().sum()) == 0.0
        D.barrier()
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        q.put((rank, repr(e)))
        raise
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


def test_single_process_defaults():
    import lanegcn_amd  # noqa: F401
    from lanegcn_amd import dist as D
    assert D.max_over_ranks(2.5) == 2.5
    assert D.job_throughput(64, 2.0) == 32.0
    assert D.shard(10, 0, 1, shuffle=False) == list(range(10))
