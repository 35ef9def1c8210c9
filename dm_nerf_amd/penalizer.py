"""Emptiness penalizer of the training step with the reference's names and signatures
(networks/penalizer.py:5-62), as fused HIP kernels: one pass over ``raw[..., 4:]`` for the loss, one for its
gradient; no ``[N,S,ins_num+1]`` temporaries."""
from __future__ import annotations

import torch

from . import ops
from ._lib import check, current_stream, load, ptr


class _PenalizerFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, raw, z, depth, rays_d, tolerance, deta_w, reduce):
        n, s, c = raw.shape
        raw = ops._dev(raw, "raw").contiguous()
        zz, stride = ops._rows(z, n, "z_vals")
        depth = ops._dev(depth, "depth").reshape(n).contiguous()
        rays_d = ops._dev(rays_d, "rays_d").contiguous()
        sums = torch.empty(4, dtype=torch.float64, device=raw.device)
        check(load().dmn_penalizer_fwd(ptr(raw), ptr(zz), stride, ptr(depth), ptr(rays_d), n, s, c, tolerance, deta_w,
                                       ptr(sums), current_stream()), "dmn_penalizer_fwd")
        if reduce is not None:              # ray-sharded training: the two loss sums and the two mask counts of ALL shards
            sums = reduce("penalizer_sums", sums)          # (penalizer.py:42-43,52 normalise by the batch's mask sums)
        ctx.save_for_backward(raw, zz, depth, rays_d, sums)
        ctx.meta = (stride, float(tolerance), float(deta_w))
        loss = torch.empty(1, dtype=torch.float32, device=raw.device)
        check(load().dmn_penalizer_loss(ptr(sums), c, ptr(loss), current_stream()), "dmn_penalizer_loss")
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g_loss):
        raw, zz, depth, rays_d, sums = ctx.saved_tensors
        stride, tolerance, deta_w = ctx.meta
        n, s, c = raw.shape
        g_raw = torch.empty_like(raw)
        g = g_loss.reshape(1).float().contiguous()
        check(load().dmn_penalizer_bwd(ptr(raw), ptr(zz), stride, ptr(depth), ptr(rays_d), n, s, c, tolerance, deta_w,
                                       ptr(sums), ptr(g), ptr(g_raw), current_stream()), "dmn_penalizer_bwd")
        return g_raw, None, None, None, None, None, None


def emptiness_penalizer(raw, z_vals, depths, rays_d, tolerance, deta_w, reduce=None):
    """``emptiness_penalizer(raw, z_vals, depths[N,1], rays_d, tolerance, deta_w)`` (networks/penalizer.py:5-55):
    scalar loss; gradient flows to ``raw[..., 4:]`` only (``depths`` is used as a constant, as the reference's
    caller detaches it).  ``reduce`` (ray-sharded training, ``parallel.ShardReducer``): sums the kernel's four
    partial sums over all shards, so that value and normalisers are the whole batch's; the gradient reaches this
    shard's samples only (sum gradients over ranks)."""
    if raw.dim() != 3 or raw.shape[-1] < 5:
        raise ValueError("raw must be [N, S, 4 + ins_num + 1]")
    if z_vals.shape[1] != raw.shape[1]:
        raise ValueError("z_vals and raw disagree on the number of samples")
    return _PenalizerFn.apply(raw, z_vals, depths.detach(), rays_d, float(tolerance), float(deta_w), reduce)


def ins_penalizer(raw, z_vals, depth, rays_d, args, reduce=None):
    """``ins_penalizer(raw, z_vals, depth[N], rays_d, args)`` with ``args.tolerance`` / ``args.deta_w``
    (networks/penalizer.py:58-62)."""
    return emptiness_penalizer(raw, z_vals, depth[..., None].detach(), rays_d, args.tolerance, args.deta_w, reduce)
