"""Multi-GPU plumbing for the render path: one process per GPU, rays (or whole images) sharded
across ranks with NO data-path collective -- rays are independent.  The only
collectives are the optional gather of the finished image and the benchmark's timing reduction,
both through ``torch.distributed`` (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the
CPU tests)."""
from __future__ import annotations

from typing import List, Tuple

import torch


def ray_shard(n_rays: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block ``[lo, hi)`` of the flat ray range owned by ``rank``; blocks differ in size
    by at most one ray and tile the range exactly."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of {world}")
    base, extra = divmod(n_rays, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def pose_shard(n_poses: int, rank: int, world: int) -> List[int]:
    """Image-sharded rendering (BASELINE config 5): pose indices rendered by ``rank``."""
    return list(range(rank, n_poses, world))


def gather_rays(local: torch.Tensor, n_rays: int, group=None) -> torch.Tensor:
    """All-gather per-rank ray blocks (made with ``ray_shard``) back into ``[n_rays, ...]``.  Equal blocks go through
    ONE ``all_gather_into_tensor`` straight into the result (RCCL: one ring pass, no concatenation); ragged ones are
    padded to the widest block.  Under the gloo backend (CPU tests, several ranks rehearsing on one GPU) device tensors
    are staged through the host, which gloo's all-gather needs."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    sizes = [ray_shard(n_rays, r, world) for r in range(world)]
    lo, hi = sizes[dist.get_rank(group)]
    if local.shape[0] != hi - lo:
        raise ValueError(f"this rank owns rays [{lo}, {hi}) but was handed {local.shape[0]}")
    staged = local.is_cuda and dist.get_backend(group) == "gloo"
    src = local.cpu() if staged else local.contiguous()
    if n_rays % world == 0:
        out = src.new_empty((n_rays,) + tuple(src.shape[1:]))
        if dist.get_backend(group) == "nccl":
            dist.all_gather_into_tensor(out, src, group=group)
        else:
            dist.all_gather(list(out.chunk(world, 0)), src, group=group)
    else:
        width = max(b - a for a, b in sizes)
        padded = src.new_zeros((width,) + tuple(src.shape[1:]))
        padded[: src.shape[0]] = src
        parts = [torch.empty_like(padded) for _ in range(world)]
        dist.all_gather(parts, padded, group=group)
        out = torch.cat([p[: b - a] for p, (a, b) in zip(parts, sizes)], 0)
    return out.to(local.device) if staged else out


def max_over_ranks(value: float, device, group=None) -> float:
    """MAX all-reduce of a host scalar (the benchmark's elapsed time)."""
    import torch.distributed as dist
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())


class GradBucket:
    """One persistent flat fp32 buffer holding every gradient of the given modules (2 x 696,338 floats = 5.57 MB for
    the coarse + fine pair).  ``DM_NeRF`` modules adopt their slice as the destination of the
    weight-gradient kernel, so after ``backward()`` the parameters' ``.grad`` already ARE views of this buffer and the
    data-parallel step is ONE in-place all-reduce over it (RCCL over xGMI: a ring is per-link bound, one large message
    is the cheap shape) -- no concatenation, no copy back.  Gradients that arrived some other way (another module
    type, a second backward into existing grads) are copied in once and re-pointed."""

    def __init__(self, models):
        self.models = list(models)
        self.params = [p for m in self.models for p in m.parameters()]
        if not self.params:
            raise ValueError("no parameters")
        dev = self.params[0].device
        self.flat = torch.zeros(sum(p.numel() for p in self.params), dtype=torch.float32, device=dev)
        self.views, offset = [], 0
        for p in self.params:
            self.views.append(self.flat[offset:offset + p.numel()].view_as(p))
            offset += p.numel()
        offset = 0
        for m in self.models:
            n = sum(p.numel() for p in m.parameters())
            if hasattr(m, "adopt_grad_buffer"):
                m.adopt_grad_buffer(self.flat[offset:offset + n])
            offset += n

    def _collect(self) -> None:
        for p, v in zip(self.params, self.views):
            if p.grad is None:
                v.zero_()
            elif p.grad.data_ptr() != v.data_ptr():
                v.copy_(p.grad)
            p.grad = v
        for m in self.models:       # whatever was lent to a backward node has been consumed by now (a backward pass that
            if hasattr(m, "_grad_lent"):      # raised half way never ran the engine callback that returns the buffer)
                m._grad_lent = False

    def allreduce(self, group=None, scale: float = 1.0) -> int:
        """SUM over ranks, then ``* scale`` (1/world for replicated-mean losses; 1.0 when every rank's loss terms are
        already weighted by its share of the batch, ``trainer.train_step(shard=...)``).  Returns the float count."""
        import torch.distributed as dist
        self._collect()
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
        if scale != 1.0:
            self.flat.mul_(scale)
        return self.flat.numel()


class _AllReduceSum(torch.autograd.Function):
    """y = sum over ranks of x.  Every rank then evaluates the same (replicated) loss on y, and its backward reaches
    only its own x: d loss / d x_r = d loss / d y, the identity -- the per-rank parameter gradients are partial sums
    that the gradient all-reduce (SUM) completes."""

    @staticmethod
    def forward(ctx, x, group):
        import torch.distributed as dist
        y = x.detach().clone()
        dist.all_reduce(y, op=dist.ReduceOp.SUM, group=group)
        return y

    @staticmethod
    def backward(ctx, g):
        return g, None


class ShardReducer:
    """``reduce(tag, tensor)`` hook of the loss heads for ray-sharded training (``ins_criterion(..., reduce=)``,
    ``ins_penalizer(..., reduce=)``): sums the batch statistics the losses match and normalise on -- the ``[3, V, K]``
    class statistics, class / ray counts, the penalizer's four sums -- over all ranks, so that an N-rank step optimises
    the loss of the whole batch (networks/evaluator.py:21,60-70, networks/penalizer.py:42-43,52).
    A few hundred bytes per call."""

    def __init__(self, group=None):
        self.group = group

    def __call__(self, tag: str, t: torch.Tensor) -> torch.Tensor:
        return _AllReduceSum.apply(t, self.group)


_buckets = {}


def allreduce_gradients(models, group=None, average: bool = True) -> int:
    """Data-parallel training step, communication part: ONE all-reduce over the persistent flat buffer of a
    ``GradBucket`` (made on first use for this set of modules).  ``average=False`` leaves the sum (shard-weighted
    losses).  Returns the number of floats reduced."""
    import torch.distributed as dist
    models = list(models)
    key = tuple(id(m) for m in models)
    bucket = _buckets.get(key)
    if bucket is None or any(a is not b for a, b in zip(bucket.params, [p for m in models for p in m.parameters()])):
        bucket = _buckets[key] = GradBucket(models)
    return bucket.allreduce(group, 1.0 / dist.get_world_size(group) if average else 1.0)
