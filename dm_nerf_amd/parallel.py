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
    """All-gather per-rank ray blocks (made with ``ray_shard``) back into ``[n_rays, ...]``."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    sizes = [ray_shard(n_rays, r, world) for r in range(world)]
    width = max(hi - lo for lo, hi in sizes)
    padded = local.new_zeros((width,) + tuple(local.shape[1:]))
    padded[: local.shape[0]] = local
    parts = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded, group=group)
    return torch.cat([p[: hi - lo] for p, (lo, hi) in zip(parts, sizes)], 0)


def max_over_ranks(value: float, device, group=None) -> float:
    """MAX all-reduce of a host scalar (the benchmark's elapsed time)."""
    import torch.distributed as dist
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())


def allreduce_gradients(models, group=None, average: bool = True) -> int:
    """Data-parallel training step, communication part: ONE all-reduce over a single flat fp32 buffer holding
    every gradient of the given modules (2 x 696,338 floats = 5.57 MB for the coarse + fine pair; SURVEY.md
    section 8e), instead of 60 small ones.  Over xGMI a ring all-reduce is per-link bound, so one large message
    is the cheap shape.  Returns the number of floats reduced."""
    import torch.distributed as dist
    params = [p for m in models for p in m.parameters() if p.grad is not None]
    if not params:
        return 0
    flat = torch.cat([p.grad.reshape(-1) for p in params])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    if average:
        flat /= dist.get_world_size(group)
    offset = 0
    for p in params:
        n = p.numel()
        p.grad.copy_(flat[offset:offset + n].view_as(p.grad))
        offset += n
    return offset
