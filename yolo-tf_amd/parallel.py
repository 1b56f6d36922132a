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


def init_from_env(backend=None, even_alone=False):
    """Initialises torch.distributed from RANK / WORLD_SIZE / LOCAL_RANK /
    MASTER_ADDR / MASTER_PORT when WORLD_SIZE > 1 (even_alone: also a one-rank
    group, so that the collective code runs on a single GPU).  Returns (rank,
    world, local_rank).  backend defaults to nccl (RCCL) when a GPU is present."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if even_alone and world == 1:
        os.environ.setdefault('MASTER_PORT', '29741')
    if (world > 1 or even_alone) and not dist.is_initialized():
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


def global_mean_loss(local_sum, local_batch=None, local_mean=None):
    """reduce_mean over the global batch.  local_sum: the dict DetectionHead.loss() returned, or
    a tensor [1] (or 0-dim) holding this rank's sum of per-image losses together with
    local_batch; local_mean: the same sum already divided by local_batch in fp32 (the loss
    kernel writes both), used as it is when there is one rank.  Returns a 0-dim float32 tensor
    on the sum's device, identical on every rank (one all-reduce of two fp64 values through a
    per-device LossReducer)."""
    dev = (local_sum['sum'] if isinstance(local_sum, dict) else local_sum).device
    reducer = _reducers.get(dev)
    if reducer is None:
        reducer = _reducers[dev] = LossReducer(dev)
    res = reducer.global_mean(local_sum, local_batch, local_mean)
    return res.tensor() if isinstance(res, PendingMean) else res


_reducers = {}


class PendingMean:
    """The global mean loss of one step while its all-reduce may still be in flight: tensor() /
    item() / float() wait for it (a stream-level wait under RCCL, no host synchronisation before
    item()).  A read never issues a collective (a rank that logs while the others do not would
    otherwise enter an all-reduce alone): with the reducer's default of one collective per step the
    value's collective has been issued before this object exists; with `batch` > 1 every rank
    calls `reducer.flush()` at the same point of its program before anybody reads, and a read of
    a step that is still waiting raises.  The value lives in one of the reducer's `depth`
    persistent slots: read it before `depth` further steps have been reduced, later reads raise."""

    def __init__(self, reducer, slot, generation):
        self._reducer, self._slot, self._generation = reducer, slot, generation
        self._done = False

    def tensor(self):
        r, k = self._reducer, self._slot
        if r._generation[k] != self._generation:
            raise RuntimeError('this step\'s mean loss was not read within %d steps: its buffer has been reused' % r.depth)
        if not self._done:
            if not r._issued[k]:
                raise RuntimeError('this step\'s pair has not been all-reduced yet: LossReducer(batch=%d) gathers steps, '
                                   'call flush() on every rank before any rank reads' % r._batch)
            if r._work[k] is not None:
                r._work[k].wait()                            # shared by the slots of one collective: waiting again is free
            r._out[k].copy_(r._buf[k, 0] / r._buf[k, 1])     # fp64 quotient, rounded to fp32 once
            self._done = True
        return r._out[k]

    def item(self):
        return float(self.tensor().item())

    __float__ = item


class LossReducer:
    """reduce_mean over the global batch (YOLOv3.py:311) for a step loop, without per-step
    allocations or host-to-device scalars: the loss kernel leaves the fp64 pair (sum of the
    per-image losses, image count) at the head of its workspace (DetectionHead.loss()
    ['reduce_pair']); that pair is copied into the next slot of a persistent ring, so the next
    step's kernels may overwrite the workspace, and the slots gathered so far are all-reduced in
    place by ONE asynchronous collective when `batch` of them are waiting, when the ring wraps, or
    on flush().  `batch` defaults to 1: one collective per step, issued by the call itself on
    every rank, so a value can be read by any subset of ranks at any time (train_one_epoch prints
    every step's).  A collective costs tens of microseconds of host time whatever its size (one
    rank, RCCL: 66 us per step with one per step against 27 us for the loss alone) and a step's
    scalar is 16 bytes, so a loop that reads the loss rarely may pass batch=n: one collective per n
    steps, and the loop calls flush() on every rank before it reads.  With one rank nothing is
    exchanged and the kernel's own fp32 mean is returned."""

    def __init__(self, device, depth=64, always_reduce=False, batch=1):
        self.depth = int(depth)
        self.always_reduce = always_reduce                   # all-reduce even in a one-rank process group
        self._batch = max(1, min(self.depth, int(batch) if batch else 1))
        self._buf = torch.zeros((self.depth, 2), dtype=torch.float64, device=device)
        self._out = torch.zeros((self.depth,), dtype=torch.float32, device=device)
        self._work = [None] * self.depth
        self._issued = [True] * self.depth
        self._generation = [0] * self.depth
        self._i = 0
        self._first, self._waiting = 0, 0                    # slots [first, first + waiting) hold pairs not yet reduced

    def flush(self):
        """Issues the collective of the slots gathered so far; every rank calls it at the same point
        of its program.  Nothing to do with the default of one collective per step."""
        n = self._waiting
        if n == 0:
            return
        lo = self._first
        part = self._buf[lo:lo + n]
        # asynchronous under RCCL, where waiting is a stream-level dependency; the CPU-side backends (gloo:
        # tests and rehearsals) complete the collective before returning
        if dist.get_backend() == 'nccl':
            work = dist.all_reduce(part, op=dist.ReduceOp.SUM, async_op=True)
        else:
            host = part.cpu()                                # gloo reduces host memory; its device path is not relied upon
            dist.all_reduce(host, op=dist.ReduceOp.SUM)
            part.copy_(host)
            work = None
        for k in range(lo, lo + n):
            self._work[k] = work
            self._issued[k] = True
        self._waiting = 0

    def global_mean(self, out_or_sum, local_batch=None, local_mean=None):
        """out_or_sum: the dict DetectionHead.loss() returned (preferred: carries reduce_pair), or
        the [1] tensor of this rank's loss sum together with local_batch (and local_mean).
        Returns a 0-dim float32 tensor with one rank, a PendingMean otherwise."""
        multi = dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or self.always_reduce)
        pair = out_or_sum.get('reduce_pair') if isinstance(out_or_sum, dict) else None
        if isinstance(out_or_sum, dict):
            local_mean = out_or_sum['mean'] if local_mean is None else local_mean
            local_sum = out_or_sum['sum']
            local_batch = int(out_or_sum['per_image'].shape[0]) if local_batch is None else local_batch
        else:
            local_sum = out_or_sum
        if not multi:
            if local_mean is not None:
                return local_mean.reshape(())
            return local_sum.reshape(()) / float(local_batch)
        k = self._i % self.depth
        self._i += 1
        if self._work[k] is not None:
            self._work[k].wait()                             # the slot's previous collective, `depth` steps ago
            self._work[k] = None
        buf = self._buf[k]
        if pair is not None:
            buf.copy_(pair)
        else:
            buf[0:1].copy_(local_sum.reshape(1))
            buf[1].fill_(float(local_batch))
        if self._waiting == 0:
            self._first = k
        self._waiting += 1
        self._issued[k] = False
        self._generation[k] += 1
        pending = PendingMean(self, k, self._generation[k])
        if self._waiting >= self._batch or k == self.depth - 1:   # a collective takes one contiguous run of slots
            self.flush()
        return pending
