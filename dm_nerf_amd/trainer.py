"""Minimal training harness around the HIP render path: what the reference's ``train()`` does per iteration
(train_dmsr.py:17-88), with the dataset / logging / flag plumbing left out (out of scope).

    optimizer = make_optimizer(model_coarse, model_fine, lrate=5e-4)            # train_dmsr.py:127-128
    losses = train_step(i, batch_rays, target_rgb, nerf, z_vals, optimizer, lrate=5e-4, lrate_decay=500,
                        ins_loss=lambda pred: dm_nerf_amd.ins_criterion(pred, target_i, ins_num)[0],
                        penalizer=lambda raw, z, depth: dm_nerf_amd.ins_penalizer(raw, z, depth, rays_d, args))

The emptiness penalizer (``networks/penalizer.py``) is ``dm_nerf_amd.ins_penalizer`` and the instance-matching loss
(``networks/evaluator.py:19-74``) is ``dm_nerf_amd.ins_criterion`` (HIP kernels for the reductions and gradients,
scipy Hungarian on the host as in the reference), both with the reference's signatures; the reference's own
functions plug in just as well.  Without them the step optimises the two colour terms only."""
from __future__ import annotations

import math
from typing import Callable, Dict, Optional, Tuple

import torch

from .render import dm_nerf


class _MseFn(torch.autograd.Function):
    """``mean((x - target) ** 2)`` (img2mse, evaluator.py:11) with its backward as one expression: the step's loss heads sit
    in a host-bound stretch of the step, every small launch there is exposed."""

    @staticmethod
    def forward(ctx, x, target):
        diff = x - target
        ctx.save_for_backward(diff)
        return torch.mean(diff * diff)

    @staticmethod
    def backward(ctx, g):
        (diff,) = ctx.saved_tensors
        return diff * (g * (2.0 / diff.numel())), None


def instance_loss(gt_labels, ins_num: int, reduce=None, assign: str = "host") -> Callable:
    """The ``ins_loss`` callable of ``train_step`` for ``dm_nerf_amd.ins_criterion`` on fixed labels: ``f(pred)`` is the loss
    sum of one map, ``f.many([coarse, fine])`` evaluates both behind one wait for the device -- or, with
    ``assign="device"`` (the Hungarian step in a kernel, ``ins_criterion``), behind none."""
    from .evaluator import ins_criterion, ins_criterion_many

    def one(pred):
        return ins_criterion(pred, gt_labels, ins_num, reduce=reduce, assign=assign)[0]

    one.many = lambda preds: [t[0] for t in ins_criterion_many(preds, gt_labels, ins_num, reduce=reduce, assign=assign)]
    return one


def make_optimizer(model_coarse, model_fine, lrate: float = 5e-4, fused: bool = False) -> torch.optim.Optimizer:
    """``Adam(coarse.parameters() + fine.parameters(), lr, betas=(0.9, 0.999))`` (train_dmsr.py:127-128).  ``fused=True``
    selects torch's single-kernel Adam (one launch instead of nine, 0.2 ms of a step): the same update up to one unit in
    the last place per step (measured), therefore not the default."""
    return torch.optim.Adam(list(model_coarse.parameters()) + list(model_fine.parameters()), lr=lrate,
                            betas=(0.9, 0.999), **({"fused": True} if fused else {}))


def decayed_lrate(i: int, lrate: float, lrate_decay: int) -> float:
    """``lrate * 0.1 ** (i / (lrate_decay * 1000))`` (train_dmsr.py:68-70)."""
    return lrate * (0.1 ** (i / (lrate_decay * 1000)))


def train_step(i: int, batch_rays, target_rgb, nerf, z_vals_coarse, optimizer, lrate: float = 5e-4,
               lrate_decay: int = 500, ins_loss: Optional[Callable] = None, penalizer: Optional[Callable] = None,
               allreduce: Optional[Callable] = None, shard: Optional[Tuple[int, int]] = None,
               report: bool = True) -> Dict[str, float]:
    """One optimisation step (train_dmsr.py:32-72).  ``nerf`` is the tuple ``create_nerf`` returns.
    ``allreduce(models)`` is called between backward and the optimizer step (data-parallel training).

    Ray-sharded, batch-exact step: pass ``shard=(rays of this rank, rays of the whole batch)``,
    loss heads built with ``reduce=parallel.ShardReducer(group)`` and ``allreduce=lambda m: bucket.allreduce(group)``
    (a SUM): the colour terms are weighted by the shard's share of the batch, the instance / emptiness terms see the
    whole batch's statistics, so the summed gradient is that of the unsharded loss for shards of any sizes.

    ``ins_loss(pred) -> loss`` is called on the coarse and on the fine instance map (train_dmsr.py:38-45); if it has a
    ``many`` attribute (``trainer.instance_loss`` builds such callables), ``ins_loss.many([coarse, fine])`` is called
    instead and evaluates both behind one wait for the device.

    Returns the loss terms as floats (one device read-back), or with ``report=False`` as detached 0-dim tensors without
    waiting for the device."""
    position_embedder, view_embedder, model_coarse, model_fine, args = nerf
    out = dm_nerf(batch_rays, position_embedder, view_embedder, model_coarse, model_fine, z_vals_coarse, args)
    mse_c = _MseFn.apply(out["rgb_coarse"], target_rgb)                             # img2mse, evaluator.py:11
    mse_f = _MseFn.apply(out["rgb_fine"], target_rgb)
    share = 1.0 if shard is None else shard[0] / shard[1]          # mean over the batch = sum of share * mean over shard
    total = (mse_f + mse_c) * share
    terms = {"rgb_coarse": mse_c, "rgb_fine": mse_f}
    empty = None
    if penalizer is not None:       # issued first: its kernels run while the host waits for the instance head's cost matrices
        empty = (penalizer(out["raw_coarse"], out["z_vals_coarse"], out["depth_coarse"]),
                 penalizer(out["raw_fine"], out["z_vals_fine"], out["depth_fine"]))
    if ins_loss is not None:                                                        # train_dmsr.py:38-45
        many = getattr(ins_loss, "many", None)       # both maps behind one wait for the device (evaluator.ins_criterion_many)
        if many is not None:
            terms["ins_coarse"], terms["ins_fine"] = many([out["ins_coarse"], out["ins_fine"]])
        else:
            terms["ins_coarse"], terms["ins_fine"] = ins_loss(out["ins_coarse"]), ins_loss(out["ins_fine"])
        total = total + terms["ins_coarse"] + terms["ins_fine"]
    if empty is not None:                                                           # train_dmsr.py:53-60 (summed in its order)
        terms["empty_coarse"], terms["empty_fine"] = empty
        total = total + terms["empty_coarse"] + terms["empty_fine"]
    optimizer.zero_grad()
    total.backward()
    if allreduce is not None:
        allreduce([model_coarse, model_fine])
    optimizer.step()
    for group in optimizer.param_groups:                                            # applied after the step, :68-72
        group["lr"] = decayed_lrate(i, lrate, lrate_decay)
    if not report:      # no read-back: the device keeps running into the next step (the reference reads its losses only
        return {**{k: v.detach() for k, v in terms.items()}, "total": total.detach()}      # every i_print steps, :74-80)
    names = list(terms) + ["total"]                                 # one read-back for all of them
    vals = torch.stack([v.detach().float().reshape(()) for v in terms.values()] + [total.detach().float()]).tolist()
    report = dict(zip(names, vals))
    report["psnr_fine"] = -10.0 * math.log10(report["rgb_fine"]) if report["rgb_fine"] > 0 else float("inf")
    return report


def checkpoint_dict(i: int, model_coarse, model_fine, optimizer) -> dict:
    """The reference's checkpoint layout (train_dmsr.py:82-87)."""
    return {"iteration": i, "network_coarse_state_dict": model_coarse.state_dict(),
            "network_fine_state_dict": model_fine.state_dict(), "optimizer_state_dict": optimizer.state_dict()}


def load_checkpoint(ckpt: dict, model_coarse, model_fine, optimizer=None) -> int:
    """Counterpart of test_dmsr.py:93-109 (the reference's training scripts never resume; this one can)."""
    model_coarse.load_state_dict(ckpt["network_coarse_state_dict"])
    model_fine.load_state_dict(ckpt["network_fine_state_dict"])
    if optimizer is not None and "optimizer_state_dict" in ckpt:
        optimizer.load_state_dict(ckpt["optimizer_state_dict"])
    return int(ckpt.get("iteration", 0))
