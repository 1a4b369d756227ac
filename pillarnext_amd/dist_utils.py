"""Data-parallel plumbing (one process per GPU, torch.distributed; backend "nccl" is RCCL on ROCm, "gloo" on CPU).

Mirrors the reference's only parallelism: frames sharded by rank (det3d/datasets/loader/build_loader.py:9-13),
SyncBatchNorm conversion + DistributedDataParallel wrap (tools/train.py:55-60), result gather by all_gather_object
(trainer/trainer/trainer.py:161-164).  Inference needs no collective on the hot path (replicas only)."""
import os

import torch
import torch.distributed as dist


def init(backend=None):
    """env:// rendezvous as torchrun sets it up (tools/train.py:26-31)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend or ("nccl" if torch.cuda.is_available() else "gloo"), init_method="env://")
    return int(os.environ.get("RANK", "0")), world, int(os.environ.get("LOCAL_RANK", "0"))


def all_reduce_sum(t, group=None):
    """In-place SUM all-reduce that also works when the backend cannot take device tensors (gloo with ROCm tensors, the
    one-GPU world-2 tests): such tensors are staged through the host."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) <= 1:
        return t
    if t.is_cuda and dist.get_backend(group) == "gloo":
        h = t.detach().cpu()
        dist.all_reduce(h, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, group=group)
    return t


class _AllReduceSumFn(torch.autograd.Function):
    """Differentiable SUM all-reduce (what torch.distributed.nn.functional.all_reduce is), on top of all_reduce_sum."""

    @staticmethod
    def forward(ctx, t, group):
        ctx.group = group
        return all_reduce_sum(t.clone(), group)

    @staticmethod
    def backward(ctx, g):
        return all_reduce_sum(g.clone(), ctx.group), None


def all_reduce_sum_autograd(t, group=None):
    return _AllReduceSumFn.apply(t, group)


def shard_frames(n_frames, rank, world):
    """DistributedSampler(shuffle=False) semantics: rank r takes r, r+world, ...; padded by wrap-around so all ranks get equally many."""
    per = (n_frames + world - 1) // world
    idx = [(rank + i * world) % max(n_frames, 1) for i in range(per)]
    return idx


def max_over_ranks(seconds, device="cpu"):
    if not (dist.is_available() and dist.is_initialized()):
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def wrap_ddp(model, device_ids=None, sync_batchnorm=True):
    """tools/train.py:55-60: convert every BatchNorm to its synchronised form, then DDP (bucketed grad all-reduce overlapped with backward)."""
    if not (dist.is_available() and dist.is_initialized()):
        return model
    if sync_batchnorm:
        from .models import convert_sync_batchnorm

        # masked BN -> global active-site statistics; every plain BN -> SyncBatchNorm (a CPU/gloo run keeps them, with a warning)
        model = convert_sync_batchnorm(model, cpu_ok=dist.get_backend() == "gloo")
    return torch.nn.parallel.DistributedDataParallel(model, device_ids=device_ids, find_unused_parameters=False)


def gather_detections(local):
    """trainer.py:161-164: every rank receives every rank's {token: detection} dict."""
    if not (dist.is_available() and dist.is_initialized()):
        return dict(local)
    dist.barrier()
    parts = [None] * dist.get_world_size()
    dist.all_gather_object(parts, local)
    out = {}
    for p in parts:
        out.update(p)
    return out
