"""Data parallelism of the path: images are sharded across ranks (one process
per GPU) and the only exchange is the loss scalar.

The reference is single-process; its batch semantics are reduce_mean over the
images of the batch (YOLOv3.py:311, YOLOv2.py:174), reproduced here over the
global batch with one all-reduce of (sum of per-image losses, image count)."""
import os

import torch
import torch.distributed as dist


def shard_range(total, rank, world_size):
    """Contiguous image range [begin, end) of `rank`; the first total % world
    ranks take one extra image."""
    base, extra = divmod(int(total), int(world_size))
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def init_from_env(backend=None):
    """Initialises torch.distributed from RANK / WORLD_SIZE / LOCAL_RANK /
    MASTER_ADDR / MASTER_PORT when WORLD_SIZE > 1.  Returns (rank, world,
    local_rank).  backend defaults to nccl (RCCL) when a GPU is present."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        if backend == 'nccl':
            torch.cuda.set_device(local_rank % max(1, torch.cuda.device_count()))
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local_rank


def global_batch(local_batch):
    """Images of the step over all ranks (every rank holds local_batch of them)."""
    if dist.is_available() and dist.is_initialized():
        return int(local_batch) * dist.get_world_size()
    return int(local_batch)


def global_mean_loss(local_sum, local_batch, local_mean=None):
    """reduce_mean over the global batch.  local_sum: tensor [1] (or 0-dim)
    holding this rank's sum of per-image losses; local_mean: the same sum
    already divided by local_batch in fp32 (the loss kernel writes both), used
    as it is when there is one rank.  Returns a 0-dim float32 tensor on
    local_sum's device, identical on every rank."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        buf = torch.empty(2, dtype=torch.float64, device=local_sum.device)
        buf[0] = local_sum.reshape(())
        buf[1] = float(local_batch)
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)
        return (buf[0] / buf[1]).to(torch.float32)
    if local_mean is not None:
        return local_mean.reshape(())                           # one rank: the fp32 mean, taken in the loss kernel
    return local_sum.reshape(()) / float(local_batch)          # one rank: the fp32 mean the reference computes
