"""Data-parallel evaluation of the flow: one process per GPU, batch sharded, ONE collective.

The reference is single-device (contextflow/model.py:168-170).  Every sample's log-density depends
only on that sample and the replicated parameters, so ranks never exchange
activations; the only exchange is an all-reduce (RCCL over xGMI on MI355X, `backend="nccl"`; gloo
in the CPU tests) of two fp64 scalars [sum_b logsumexp_m logp[b, :], number of samples] from which
the global mean NLL / bits-per-dim follows.  The message is 16 bytes: latency-bound, not link-bound.
"""
import math
import os

import torch
import torch.distributed as dist


def _active():
    """Collectives run when a process group of more than one rank exists - or of ONE rank with CF_DIST_SINGLE_RANK=1
    (tests on a one-GPU box: the same calls through a real RCCL communicator of size 1)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or os.environ.get("CF_DIST_SINGLE_RANK") == "1"


def env_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init_process_group(backend=None):
    """Initialise torch.distributed from the torchrun environment (no-op for world size 1)."""
    rank, local_rank, world = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            # one process per GPU: every local rank needs its own device (RCCL deadlocks or fails obscurely otherwise)
            n = torch.cuda.device_count()
            if local_rank >= n:
                raise RuntimeError("contextflow_amd.dist: LOCAL_RANK %d but only %d visible GPU(s) (WORLD_SIZE %d): launch one "
                                   "process per GPU, e.g. torch.distributed.run --nproc-per-node <= %d" % (local_rank, n, world, n))
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
        if dist.get_world_size() != world:
            raise RuntimeError("contextflow_amd.dist: process group of %d ranks, WORLD_SIZE says %d" % (dist.get_world_size(), world))
    return rank, local_rank, world


def shard_bounds(total, rank, world):
    """Contiguous, balanced shard [lo, hi) of `total` samples for `rank` (first `total % world` ranks get one more)."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def allreduce_nll(local_sum_logp, local_count):
    """All-reduce [sum of per-sample log p, sample count] (fp64).  `local_sum_logp` may be a 0-dim/1-elem
    tensor on the compute device.  Returns the reduced (sum, count) as a 2-element fp64 tensor."""
    dev = local_sum_logp.device if torch.is_tensor(local_sum_logp) else torch.device("cpu")
    buf = torch.zeros(2, dtype=torch.float64, device=dev)
    buf[0] = local_sum_logp.reshape(-1)[0].double() if torch.is_tensor(local_sum_logp) else float(local_sum_logp)
    buf[1] = float(local_count)
    if _active():
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    return buf


def mean_bits_per_dim(reduced, dims):
    """bits/dim = -(sum log p / count) / (D ln 2)."""
    return float(-(reduced[0] / reduced[1]) / (dims * math.log(2.0)))


def broadcast_parameters(module, src=0):
    """Replicate parameters and buffers from `src` (after rank `src` ran the ActNorm data-dependent init): ONE flat
    message per dtype (the cifar10 flow: 6 MB of fp32 + 12 int64 flags) instead of one tiny broadcast per tensor, written
    back with `copy_` so that every tensor's version counter moves (parameter-derived caches key on it)."""
    if not _active():
        return
    tensors = list(module.parameters()) + list(module.buffers())
    with torch.no_grad():
        for dtype in sorted({t.dtype for t in tensors}, key=str):
            group = [t for t in tensors if t.dtype == dtype and t.numel() > 0]
            flat = torch.cat([t.detach().reshape(-1) for t in group])
            dist.broadcast(flat, src=src)
            o = 0
            for t in group:
                n = t.numel()
                t.copy_(flat[o:o + n].view_as(t))
                o += n
    for m in module.modules():
        if hasattr(m, "_init_done") and hasattr(m, "initialized"):
            m._init_done = bool(int(m.initialized.item()))
        if hasattr(m, "invalidate_caches"):
            m.invalidate_caches()


def allreduce_actnorm_sums(sums):
    """ActNorm's data-dependent init over a SHARDED batch (actnorm.py:28-35): `sums` = fp64
    [sum x (C) | sum x^2 (C) | elements per channel (1)] of this rank's shard, summed in place over the ranks - one
    (2C+1)-double message per ActNorm layer, after which every rank derives the same t / logs as a single process would
    from the whole batch (cf_actnorm_from_sums).  No-op for world size 1."""
    if _active():
        dist.all_reduce(sums, op=dist.ReduceOp.SUM)
    return sums


class sharded_actnorm_init:
    """Context manager for the first (initialising) forward of a data-parallel run: every rank passes ITS shard of the
    init batch through the model and all ranks end up with the global-batch ActNorm statistics - instead of rank 0
    initialising alone and broadcasting (`broadcast_parameters`)."""

    def __enter__(self):
        from .layers.actnorm import ActNorm
        self._prev, ActNorm.sharded_init = ActNorm.sharded_init, True
        return self

    def __exit__(self, *exc):
        from .layers.actnorm import ActNorm
        ActNorm.sharded_init = self._prev
        return False


class GradBucket:
    """Gradient storage of a data-parallel training step: ONE flat, persistent fp32 tensor that holds the
    gradients of all trainable parameters in the order the hand-written backward produces them, cut into segments at the
    resolution levels of the flow.  `view(p)` is the slice that the gradient kernels of parameter p write and that becomes
    `p.grad` - there is no concatenation and no copy back; `reduce(i)` enqueues the all-reduce of segment i (RCCL over
    xGMI; one message per level: for the cifar10 flow 3.9 MB for the 4x4 level with the final prior, 1.3 MB, 1.5 MB) the
    moment the last kernel that writes into it has been launched - RCCL runs it on its own stream behind those kernels,
    next to the backward of the levels below - and `finish()` makes the current stream (the optimizer's) wait for all of
    them and turns the sums into means.  Works under HIP-graph capture (the collectives are captured with the step).

    groups: [[parameters of segment 0], ...] in backward order; parameters listed twice keep their first slot."""

    def __init__(self, groups, device):
        self.slots, self.segments, o = {}, [], 0
        for g in groups:
            lo = o
            for p in g:
                if p not in self.slots:
                    n = p.numel()
                    self.slots[p] = (o, n)
                    o += (n + 3) & ~3                   # 16-byte aligned slots: the gradient kernels store float4s
            if o > lo:
                self.segments.append((lo, o))
        self.flat = torch.zeros(max(o, 1), device=device, dtype=torch.float32)
        self._views = {p: self.flat[lo:lo + n].view(p.shape) for p, (lo, n) in self.slots.items()}
        self._works = []
        self.key = tuple(id(p) for g in groups for p in g)

    def view(self, p):
        return self._views.get(p)

    def segment_of(self, p):
        lo = self.slots[p][0]
        return next(i for i, (a, b) in enumerate(self.segments) if a <= lo < b)

    def reduce(self, i):
        """Enqueue the all-reduce (SUM) of segment i behind whatever the current stream has been given so far."""
        if _active():
            lo, hi = self.segments[i]
            self._works.append(dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, async_op=True))

    def finish(self):
        """The current stream waits for the collectives enqueued by reduce(); sums -> means over the ranks."""
        works, self._works = self._works, []
        for w in works:
            w.wait()
        if works and dist.get_world_size() > 1:
            self.flat.mul_(1.0 / dist.get_world_size())

    def message_bytes(self):
        return [4 * (hi - lo) for lo, hi in self.segments]


def allreduce_gradients(module, bucket_bytes=32 << 20):
    """Average parameter gradients over the ranks (data-parallel training step).  Gradients are packed into flat
    fp32 buckets (default 32 MB; the whole cifar10 flow is 6 MB = one message) so that each RCCL all-reduce moves a
    bandwidth-relevant payload over the point-to-point xGMI links instead of 135 tiny latency-bound ones."""
    if not _active():
        return
    world = dist.get_world_size()
    params = [p for p in module.parameters() if p.grad is not None]
    bucket, size = [], 0

    def flush():
        nonlocal bucket, size
        if not bucket:
            return
        flat = torch.cat([p.grad.reshape(-1) for p in bucket])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat.div_(world)
        o = 0
        for p in bucket:
            n = p.grad.numel()
            p.grad.copy_(flat[o:o + n].view_as(p.grad))
            o += n
        bucket, size = [], 0

    for p in params:
        bucket.append(p)
        size += p.grad.numel() * 4
        if size >= bucket_bytes:
            flush()
    flush()
