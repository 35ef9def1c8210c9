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
    # With the Hungarian step on the device (assign="device") nothing waits for the loss head's status words, and a label
    # outside [0, ins_num) or a failed assignment shows up as non-finite instance terms with zero gradients (the host
    # form raises at once, as the reference's indexing does): this read-back is the first point that waits -- fail there.
    bad = [k for k in ("ins_coarse", "ins_fine") if k in report and not math.isfinite(report[k])]
    if bad:
        raise ValueError(f"instance-matching loss is not finite ({', '.join(bad)}): a label outside [0, ins_num) or a failed "
                         "device-side assignment (assign='device' reports neither at the call; assign='host' raises there)")
    report["psnr_fine"] = -10.0 * math.log10(report["rgb_fine"]) if report["rgb_fine"] > 0 else float("inf")
    return report


class GraphedTrainStep:
    """One optimisation step captured in a hipGraph (``torch.cuda.CUDAGraph``): forward of both networks, the loss heads
    (``ins_loss`` must be sync-free: ``instance_loss(..., assign="device")``), the hand-written backward, Adam and the
    learning-rate decay replay as ONE host call instead of ~85 kernel launches.  What ``train_step(report=False)`` does
    per iteration (train_dmsr.py:32-72), with the state that changes from step to step held in device tensors:

      step.rays [2, N, 3], step.target [N, 3]   static input buffers: fill them in place (``copy_``, or let the selector
                                                kernels write there), then call ``step()``
      step.lr, step.iteration                   learning rate (decayed in the graph AFTER the update, as :68-72) and the
                                                iteration count it is computed from
      step.terms                                the loss terms of the last replay (0-dim device tensors; read them when
                                                you would print them -- that read is the only wait for the device)

    The constructor runs ``warmup`` eager steps (they are real steps: weights move, ``iteration`` counts them) so that
    every lazily built table exists before the capture, then captures one.  The optimizer is torch's Adam with
    ``capturable=True`` and a tensor learning rate (``step.optimizer``); checkpoint it as usual.  Labels closed over by
    ``ins_loss`` are read from their tensor at every replay (their class table is rebuilt inside the graph)."""

    def __init__(self, batch_rays, target_rgb, nerf, z_vals_coarse, lrate: float = 5e-4, lrate_decay: int = 500,
                 ins_loss: Optional[Callable] = None, penalizer: Optional[Callable] = None, first_iteration: int = 0,
                 warmup: int = 2):
        position_embedder, view_embedder, model_coarse, model_fine, args = nerf
        dev = batch_rays.device
        self.rays, self.target = batch_rays.clone(), target_rgb.clone()
        self.lr = torch.tensor(float(decayed_lrate(first_iteration, lrate, lrate_decay)), device=dev)
        self.iteration = torch.tensor(float(first_iteration), device=dev)
        self.optimizer = torch.optim.Adam(list(model_coarse.parameters()) + list(model_fine.parameters()), lr=self.lr,
                                          betas=(0.9, 0.999), capturable=True)
        scale = 1.0 / (lrate_decay * 1000.0)
        tenth = torch.tensor(0.1, device=dev)

        def body():
            out = dm_nerf(self.rays, position_embedder, view_embedder, model_coarse, model_fine, z_vals_coarse, args)
            terms = {"rgb_coarse": _MseFn.apply(out["rgb_coarse"], self.target),
                     "rgb_fine": _MseFn.apply(out["rgb_fine"], self.target)}
            total = terms["rgb_fine"] + terms["rgb_coarse"]
            empty = None
            if penalizer is not None:
                empty = (penalizer(out["raw_coarse"], out["z_vals_coarse"], out["depth_coarse"]),
                         penalizer(out["raw_fine"], out["z_vals_fine"], out["depth_fine"]))
            if ins_loss is not None:
                many = getattr(ins_loss, "many", None)
                if many is not None:
                    terms["ins_coarse"], terms["ins_fine"] = many([out["ins_coarse"], out["ins_fine"]])
                else:
                    terms["ins_coarse"], terms["ins_fine"] = ins_loss(out["ins_coarse"]), ins_loss(out["ins_fine"])
                total = total + terms["ins_coarse"] + terms["ins_fine"]
            if empty is not None:
                terms["empty_coarse"], terms["empty_fine"] = empty
                total = total + terms["empty_coarse"] + terms["empty_fine"]
            self.optimizer.zero_grad(set_to_none=True)
            total.backward()
            self.optimizer.step()
            self.lr.copy_(lrate * torch.pow(tenth, self.iteration * scale))      # lrate * 0.1 ** (i / (decay * 1000)), :68-72
            self.iteration += 1.0
            return {**{k: v.detach() for k, v in terms.items()}, "total": total.detach()}

        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                body()
        torch.cuda.current_stream(dev).wait_stream(side)
        from . import evaluator
        evaluator._LABEL_PREP.clear()           # the labels' class table is rebuilt inside the graph, at every replay
        self.optimizer.zero_grad(set_to_none=True)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.terms = body()

    def __call__(self) -> Dict[str, torch.Tensor]:
        self.graph.replay()
        return self.terms


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
