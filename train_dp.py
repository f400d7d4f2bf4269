#!/usr/bin/env python
"""Data-parallel training driver with the loop semantics of the reference's train.py, on RCCL instead of
Horovod/MPI (one process per GPU: `python -m torch.distributed.run --nproc-per-node N train_dp.py`).

Kept from the reference (train.py:41-259): CLI flags -m/--model, --eval, --resume, --weight; seed = rank
(:55-59); the model plugin call import_module(model).get_model() (:63-64); parameter broadcast from rank 0
(:96,145); per-rank sampler shard with drop_last (:119-131); step order net -> loss -> post_process -> zero_grad ->
backward -> [gradient average] -> opt.step(epoch) with epoch += 1/num_batches per iteration (:175-186); checkpoint
dict {"epoch", "state_dict" (CPU tensors), "opt_state"} named "%3.3f.ckpt" (:230-242) every save_iters; display +
metrics reset every display_iters; a validation pass every val_iters and at the end (:189-208).  The gradient average
is ONE flat all-reduce (lanegcn_amd.dist.allreduce_mean_grads).  Difference: the sampler drops the shard's tail
where the reference's DistributedSampler pads it.
"""
import argparse
import os
import sys
import time
from importlib import import_module

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import lanegcn_amd  # noqa: E402,F401
from lanegcn_amd import dist as D  # noqa: E402
from lanegcn_amd.utils import load_pretrain  # noqa: E402


def save_ckpt(net, opt, save_dir, epoch):
    os.makedirs(save_dir, exist_ok=True)
    state = {k: v.cpu() for k, v in net.state_dict().items()}
    path = os.path.join(save_dir, "%3.3f.ckpt" % epoch)
    torch.save({"epoch": epoch, "state_dict": state, "opt_state": opt.opt.state_dict()}, path)
    return path


def batches(dataset, collate_fn, batch_size, rank, world, seed, epoch, shuffle=True):
    idx = D.shard(len(dataset), rank, world, seed=seed, epoch=epoch, shuffle=shuffle, drop_last=True)
    for i in range(0, len(idx) - batch_size + 1, batch_size):        # drop_last on the batch level too
        yield collate_fn([dataset[j] for j in idx[i:i + batch_size]])


def main(argv=None):
    ap = argparse.ArgumentParser(description="LaneGCN training on MI355X (RCCL data parallel)")
    ap.add_argument("-m", "--model", default="lanegcn_mi355x", type=str, metavar="MODEL", help="model plugin module")
    ap.add_argument("--eval", action="store_true")
    ap.add_argument("--resume", default="", type=str, metavar="RESUME", help="checkpoint path")
    ap.add_argument("--weight", default="", type=str, metavar="WEIGHT", help="checkpoint path (weights only)")
    ap.add_argument("--max-iters", type=int, default=0, help="stop after this many iterations (smoke runs)")
    ap.add_argument("--batch-size", type=int, default=0, help="override config['batch_size']")
    ap.add_argument("--save-dir", default="", help="override config['save_dir']")
    ap.add_argument("--dataset-len", type=int, default=0, help="synthetic dataset length override")
    args = ap.parse_args(argv)

    rank, world, local_rank = D.env_ranks()
    torch.cuda.set_device(local_rank)
    if world > 1:
        torch.distributed.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    seed = rank                                                   # train.py:55-59
    torch.manual_seed(seed)
    np.random.seed(seed)

    model = import_module(args.model)
    config, Dataset, collate_fn, net, loss, post_process, opt = model.get_model()
    if args.batch_size:
        config["batch_size"] = args.batch_size
    if args.save_dir:
        config["save_dir"] = args.save_dir
    if args.resume or args.weight:
        ckpt = torch.load(args.resume or args.weight, map_location="cpu", weights_only=True)
        load_pretrain(net, ckpt["state_dict"])
        if args.resume:
            config["epoch"] = ckpt["epoch"]
            opt.load_state_dict(ckpt["opt_state"])
    D.broadcast_parameters(net.state_dict().values(), 0)

    kw = {"length": args.dataset_len} if args.dataset_len else {}
    dataset = Dataset(config.get("train_split"), config, train=True, **kw)
    if args.eval:
        net.eval()
        with torch.no_grad():
            for data in batches(dataset, collate_fn, config["val_batch_size"], rank, world, 0, 0, shuffle=False):
                out = net(data)
                loss_out = loss(out, data)
                if rank == 0:
                    print("val loss %.4f" % float(loss_out["loss"]))
                break
        return 0

    # The reference runs its training loop on one Python thread per GPU (train.py:8-10, 175-186); with ~375 Python
    # autograd Functions in a backward, running them on the calling thread instead of the engine's device thread
    # saves 12 % of a step.  Scoped to this driver (LGCN_AUTOGRAD_MT=1 keeps torch's default).
    if os.environ.get("LGCN_AUTOGRAD_MT", "0") != "1":
        torch.autograd.set_multithreading_enabled(False)
    val_set = Dataset(config.get("val_split"), config, train=False, **kw)

    def validate(epoch):                                             # train.py:200-216
        net.eval()
        t_val, vm = time.time(), dict()
        with torch.no_grad():
            for vdata in batches(val_set, collate_fn, config["val_batch_size"], rank, world, 0, 0, shuffle=False):
                vout = net(vdata)
                post_process.append(vm, loss(vout, vdata), post_process(vout, vdata))
        vm = D.gather_metrics(vm)
        if rank == 0 and vm:
            post_process.display(vm, time.time() - t_val, epoch)
        net.train()

    net.train()
    # The shard drops the tail that does not divide by the world size and by the batch size (the reference's
    # DistributedSampler pads the shard by repeating samples, train.py:119-131; its DataLoader then drops the last
    # partial batch): every rank sees the same number of full batches either way.
    per_rank = len(dataset) // world
    num_batches = max(per_rank // config["batch_size"], 1)
    save_iters = int(np.ceil(config["save_freq"] * num_batches))                       # train.py:166-171
    display_iters = max(int(config["display_iters"] / (world * config["batch_size"])), 1)
    val_iters = max(int(config["val_iters"] / (world * config["batch_size"])), 1)
    # a resumed epoch is a float accumulated in steps of 1 / num_batches: snap it to the batch grid before int()
    epoch = round(float(config["epoch"]) * num_batches) / num_batches
    it, t0, metrics = 0, time.time(), dict()
    last_path, done = None, False
    bucket = D.GradBucket(net.parameters())                        # persistent flat gradient buffer (p.grad = views of it)
    for ep in range(int(epoch + 0.5 / num_batches), config["num_epochs"]):
        for data in batches(dataset, collate_fn, config["batch_size"], rank, world, seed=0, epoch=ep):
            epoch += 1.0 / num_batches
            out = net(data)
            loss_out = loss(out, data)
            post_out = post_process(out, data)
            post_process.append(metrics, loss_out, post_out)
            bucket.zero()                                          # the reference's opt.zero_grad(): one fill of the flat bucket
            loss_out["loss"].backward()                            # accumulates into the bucket's views
            bucket.allreduce_mean()                                # Horovod DistributedOptimizer semantics, one in-place all-reduce
            lr = opt.step(epoch)
            it += 1
            num_iters = int(np.round(epoch * num_batches))
            finished = epoch >= config["num_epochs"] or (args.max_iters and it >= args.max_iters)
            if rank == 0 and (num_iters % save_iters == 0 or finished):                 # train.py:189-193
                last_path = save_ckpt(net, opt, config["save_dir"], epoch)
            if num_iters % display_iters == 0 or (args.max_iters and it % 10 == 0):      # train.py:195-201
                metrics = D.gather_metrics(metrics)
                if rank == 0:
                    post_process.display(metrics, time.time() - t0, epoch, lr)
                t0, metrics = time.time(), dict()
            if num_iters % val_iters == 0 or finished:                                   # train.py:203-208
                validate(epoch)
            if finished:
                done = True
                break
        if done:
            break
    D.barrier()
    if rank == 0 and last_path:
        print("saved", last_path)
    if world > 1:
        torch.distributed.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
