"""Loss heads of the training step with the reference's names (networks/evaluator.py:11-74).

``ins_criterion`` matches predicted instance columns to ground-truth classes with the Hungarian algorithm on the
host (scipy, exactly as the reference does) -- but the cost matrices come from class-wise column statistics reduced
by a HIP kernel (``csrc/ins_stats.hip``) instead of three ``[ins, ins, N]`` broadcast tensors, and the gradient
flows back through the matching gather kernel."""
from __future__ import annotations

import numpy as np
import torch

from . import ops
from ._lib import check, current_stream, load, ptr

img2mse = lambda x, y: torch.mean((x - y) ** 2)                                     # evaluator.py:11
mse2psnr = lambda x: -10.0 * torch.log(x) / np.log(10.0)                            # evaluator.py:15


class _ClassStatsFn(torch.autograd.Function):
    """pred [N,K], rays grouped by class -> stats [3, V, K] (sums of log(p+1e-8), log(1-p+1e-8), p per class)."""

    @staticmethod
    def forward(ctx, pred, classes, n_classes):
        pred = ops._dev(pred, "pred_ins").contiguous()
        n, k = pred.shape
        order = torch.argsort(classes, stable=True)
        seg = torch.zeros(n_classes + 1, dtype=torch.int64, device=pred.device)
        seg[1:] = torch.cumsum(torch.bincount(classes, minlength=n_classes), 0)
        stats = torch.empty(3, n_classes, k, dtype=torch.float32, device=pred.device)
        check(load().dmn_ins_stats_fwd(ptr(pred), ptr(order), ptr(seg), n, k, n_classes, ptr(stats), current_stream()),
              "dmn_ins_stats_fwd")
        ctx.save_for_backward(pred, classes)
        ctx.n_classes = n_classes
        return stats

    @staticmethod
    def backward(ctx, g_stats):
        pred, classes = ctx.saved_tensors
        n, k = pred.shape
        g_pred = torch.empty_like(pred)
        check(load().dmn_ins_stats_bwd(ptr(pred), ptr(classes), n, k, ctx.n_classes, ptr(g_stats.contiguous()),
                                       ptr(g_pred), current_stream()), "dmn_ins_stats_bwd")
        return g_pred, None, None


# classes / counts of the last labels tensor seen (the coarse and the fine call of a step share their labels)
_LABEL_PREP = {}


def _label_prep(labels: torch.Tensor, ins_num: int):
    key = (labels.data_ptr(), labels._version, labels.shape[0], ins_num, labels.device)
    hit = _LABEL_PREP.get("last")
    if hit is not None and hit[0] == key and hit[1] is labels:
        return hit[2]
    n = labels.shape[0]
    classes = torch.empty(n, dtype=torch.int64, device=labels.device)
    count = torch.empty(ins_num, dtype=torch.float32, device=labels.device)
    meta = torch.empty(2, dtype=torch.int32, device=labels.device)
    check(load().dmn_ins_label_prep(ptr(labels), n, ins_num, ptr(classes), ptr(count), ptr(meta), current_stream()),
          "dmn_ins_label_prep")
    _LABEL_PREP["last"] = (key, labels, (classes, count, meta))
    return classes, count, meta


class _InsCriterionFn(torch.autograd.Function):
    """The whole loss head on the device, for one or several predictions that share their labels (the coarse and the fine
    map of a training step): per prediction five launches, and for all of them together ONE read-back (the cost matrices)
    and one upload (the matchings).  ``apply(labels, ins_num, *preds)`` -> (sum, valid_ce, invalid_ce, valid_siou) per
    prediction, flattened; all None for inputs the fused kernels do not take (labels outside [0, 1024), more labels than
    columns): the caller falls back to the tensor-operation path."""

    @staticmethod
    def forward(ctx, labels, ins_num, reduce, assign, *preds):
        from scipy.optimize import linear_sum_assignment
        lib = load()
        preds = [ops._dev(p, "pred_ins").contiguous() for p in preds]
        n, k = preds[0].shape
        dev, st = preds[0].device, current_stream()
        n_dev = None
        if reduce is None:
            classes, count, meta = _label_prep(labels, ins_num)
        else:       # ray-sharded: label set, class counts and ray count of ALL shards (torch.unique over the whole batch)
            classes = torch.empty(n, dtype=torch.int64, device=dev)
            count = torch.empty(ins_num, dtype=torch.float32, device=dev)
            meta = torch.empty(2, dtype=torch.int32, device=dev)
            # ins_num + 1 label bins and, last, the label-out-of-range flag: summed with the bins, it is the same on every rank
            hist = torch.empty(ins_num + 2, dtype=torch.float32, device=dev)
            check(lib.dmn_ins_label_hist(ptr(labels), n, ins_num + 1, ptr(hist), ptr(meta), st), "dmn_ins_label_hist")
            hist = reduce("ins_label_count", hist).contiguous()
            check(lib.dmn_ins_label_slots(ptr(labels), n, ptr(hist), ins_num + 1, ins_num, ptr(classes), ptr(count),
                                          ptr(meta), st), "dmn_ins_label_slots")
            n_dev = reduce("ins_ray_count", torch.full((1,), float(n), dtype=torch.float32, device=dev)).contiguous()
        m, stride = len(preds), 2 + ins_num * k
        stats = torch.empty(m, 3, ins_num, k, dtype=torch.float32, device=dev)
        partial = torch.empty(m, lib.dmn_ins_stats_chunks(n), 3, ins_num, k, dtype=torch.float64, device=dev)
        cost = torch.empty(m, 2, ins_num, k, dtype=torch.float32, device=dev)
        colsum = torch.empty(m, 2, k, dtype=torch.float32, device=dev)
        blob = torch.empty(m, stride, dtype=torch.float32, device=dev)
        for i, pred in enumerate(preds):
            check(lib.dmn_ins_stats_scan(ptr(pred), ptr(classes), n, k, ins_num, ptr(partial[i]), st), "dmn_ins_stats_scan")
            if reduce is not None:
                check(lib.dmn_ins_stats_sum(ptr(partial[i]), n, k, ins_num, ptr(stats[i]), st), "dmn_ins_stats_sum")
        if reduce is not None:      # the [3, V, K] statistics of all shards (evaluator.py:60-70 see the whole batch)
            stats = reduce("ins_class_stats", stats).contiguous()
        for i in range(m):
            check(lib.dmn_ins_cost(None if reduce is not None else ptr(partial[i]), ptr(stats[i]), ptr(count), ptr(meta), n,
                                   ptr(n_dev), k, ins_num, ptr(cost[i]), ptr(colsum[i]), ptr(blob[i]), st), "dmn_ins_cost")
        if assign == "device":          # the same assignment without the wait (csrc/ins_stats.hip: ins_assign_kernel)
            match_t = torch.empty(m, ins_num + k, dtype=torch.int32, device=dev)
            check(lib.dmn_ins_assign(ptr(blob), ptr(meta), k, ins_num, ptr(match_t), m, st), "dmn_ins_assign")
        else:
            host = blob.cpu().numpy()                                                 # the one wait for the device
            v = int(host[0, 0])
            if reduce is not None and host[0, 1] != 0.0:
                # the flag was summed with the histogram: every rank is here, none re-enters the collectives
                raise ValueError("sharded ins_criterion needs labels in [0, ins_num] on every shard (and no more distinct "
                                 "labels than instance columns)")
            if host[0, 1] != 0.0 or v == 0:
                return (None,) * (4 * m)
            match = np.full((m, ins_num + k), -1, dtype=np.int32)
            for i in range(m):
                rows, cols = linear_sum_assignment(host[i, 2:].reshape(ins_num, k)[:v])    # reorder(), :42-52
                match[i, rows] = cols                                                 # col_of [ins_num]
                match[i, ins_num + cols] = rows                                       # class_of [k]
            match_t = torch.from_numpy(match).to(dev)
        out = torch.empty(m, 4, dtype=torch.float32, device=dev)
        for i in range(m):
            check(lib.dmn_ins_loss(ptr(cost[i]), ptr(colsum[i]), ptr(match_t[i]), ptr(match_t[i]) + 4 * ins_num, ptr(meta),
                                   n, ptr(n_dev), k, ins_num, ptr(out[i]), st), "dmn_ins_loss")
        ctx.save_for_backward(classes, stats, count, colsum, match_t, meta, *preds)
        ctx.ins_num, ctx.n_dev = ins_num, n_dev
        ctx.set_materialize_grads(False)               # terms nobody differentiates arrive as None, not as zero tensors
        return tuple(out[i, j] for i in range(m) for j in range(4))      # outputs of this node: no select nodes behind them

    @staticmethod
    def backward(ctx, *grads):
        classes, stats, count, colsum, match_t, meta, *preds = ctx.saved_tensors
        lib, st = load(), current_stream()
        n, k = preds[0].shape
        g_preds = []
        for i, pred in enumerate(preds):
            g4 = grads[4 * i: 4 * i + 4]
            if all(t is None for t in g4):
                g_preds.append(None)
                continue
            g = [None if t is None else t.reshape(1).float().contiguous() for t in g4]
            g_stats = torch.empty_like(stats[i])
            check(lib.dmn_ins_loss_bwd(ptr(stats[i]), ptr(count), ptr(colsum[i]), ptr(match_t[i]) + 4 * ctx.ins_num,
                                       ptr(meta), n, ptr(ctx.n_dev), k, ctx.ins_num, ptr(g[0]), ptr(g[1]), ptr(g[2]),
                                       ptr(g[3]), ptr(g_stats), st), "dmn_ins_loss_bwd")
            g_pred = torch.empty_like(pred)
            check(lib.dmn_ins_stats_bwd(ptr(pred), ptr(classes), n, k, ctx.ins_num, ptr(g_stats), ptr(g_pred), st),
                  "dmn_ins_stats_bwd")
            g_preds.append(g_pred)
        return (None, None, None, None, *g_preds)


def ins_criterion_many(preds, gt_labels: torch.Tensor, ins_num: int, reduce=None, assign: str = "host"):
    """``ins_criterion`` for several predictions with the same labels (``[ins_coarse, ins_fine]`` of a training step,
    train_dmsr.py:38-45) behind ONE wait for the device instead of one per prediction: a list of the 4-tuples
    ``ins_criterion`` returns.  Falls back to separate calls where the fused kernels do not apply."""
    preds = list(preds)
    labels = gt_labels if gt_labels.dtype == torch.int64 else gt_labels.long()
    ok = (len(preds) > 0 and all(p.dim() == 2 and p.shape == preds[0].shape and p.is_cuda for p in preds)
          and preds[0].shape[1] == ins_num and 0 < preds[0].shape[0] == labels.shape[0] and ins_num <= 128)
    if ok:
        got = _InsCriterionFn.apply(labels.to(preds[0].device).contiguous(), ins_num, reduce, assign, *preds)
        if got[0] is not None:
            return [tuple(got[4 * i: 4 * i + 4]) for i in range(len(preds))]
    return [ins_criterion(p, gt_labels, ins_num, reduce=reduce) for p in preds]


def ins_criterion(pred_ins: torch.Tensor, gt_labels: torch.Tensor, ins_num: int, reduce=None, fused: bool = True,
                  assign: str = "host"):
    """``(ins_loss_sum, valid_ce, invalid_ce, valid_siou)`` of networks/evaluator.py:19-38 for ``pred_ins [N, ins_num]``
    (the rendered, sigmoided instance map) and integer ``gt_labels [N]``.  All four are 0-dim tensors
    (the reference's ``invalid_ce`` is ``tensor([0])`` when every column is matched).

    ``reduce`` (ray-sharded training, ``parallel.ShardReducer``): called as ``reduce(tag, tensor)`` on the label-presence
    mask, the class counts, the ray count and the ``[3, V, K]`` class statistics; it returns their sums over all shards.
    Matching and normalisation then see the whole batch (evaluator.py:21,60-70), every rank computes the same loss
    value, and its backward reaches the rank's own rays only -- so gradients must be SUMMED over ranks.

    ``assign``: "host" runs the Hungarian step with scipy on the host as the reference does (one wait for the device per
    call); "device" runs the same algorithm (scipy's shortest augmenting paths, fp64, its choice among equal candidates) in
    a kernel, so that the call never waits -- labels outside [0, 1024) or more labels than columns then give NaN losses
    instead of the fallback / the error of the host form.

    The loss head runs as fused kernels (``csrc/ins_stats.hip``: label classes, class statistics, cost matrices, loss
    terms and their gradient; under ``reduce`` with the three all-reduces in between), with the Hungarian step on the host;
    ``fused=False`` is the same arithmetic in tensor operations."""
    from scipy.optimize import linear_sum_assignment

    if pred_ins.dim() != 2 or pred_ins.shape[1] != ins_num or gt_labels.shape[0] != pred_ins.shape[0]:
        raise ValueError("pred_ins must be [N, ins_num] and gt_labels [N]")
    if assign not in ("host", "device"):
        raise ValueError("assign is 'host' or 'device'")
    n = pred_ins.shape[0]
    labels = gt_labels if gt_labels.dtype == torch.int64 else gt_labels.long()
    if fused and n > 0 and ins_num <= 128 and pred_ins.is_cuda:
        # five launches instead of ~100 tensor operations (the host is the bottleneck of this part of a training step);
        # ``fused=False`` is the same arithmetic written with tensor operations
        got = _InsCriterionFn.apply(labels.to(pred_ins.device).contiguous(), ins_num, reduce, assign, pred_ins)
        if got[0] is not None:
            return got
    if reduce is None:
        valid, classes = torch.unique(labels, return_inverse=True)                    # :21, :25 (sorted labels)
        v = int(valid.numel())                                                        # :24
        count = torch.bincount(classes, minlength=v).to(torch.float32)[:, None]
    else:
        # labels present in ANY shard, in sorted order (torch.unique over the whole batch).  A label outside [0, ins_num]
        # on one shard must stop EVERY rank before the next collective: the flag rides behind the histogram (as in the
        # kernels' form, dmn_ins_label_hist), so the summed flag is the same everywhere
        ok = (labels >= 0) & (labels <= ins_num)
        hist = torch.bincount(labels[ok], minlength=ins_num + 1).to(torch.float32)
        hist = reduce("ins_label_count", torch.cat([hist, (~ok).any().to(torch.float32).reshape(1)]))
        if float(hist[-1]) != 0.0:
            raise ValueError("sharded ins_criterion needs labels in [0, ins_num] on every shard")
        per_label = hist[:-1]
        present = per_label > 0
        v = int(present.sum())
        slot = torch.cumsum(present.long(), 0) - 1                                    # label -> row of the class axis
        classes = slot[labels]
        count = per_label[present][:, None]
        n = int(round(float(reduce("ins_ray_count", torch.tensor([float(n)], device=pred_ins.device)))))
    if v > ins_num:
        raise ValueError(f"{v} distinct labels but only {ins_num} instance columns")
    stats = _ClassStatsFn.apply(pred_ins, classes.contiguous(), v)
    if reduce is not None:
        stats = reduce("ins_class_stats", stats)
    s1, s0, tp = stats[0], stats[1], stats[2]
    cost_ce = -(s1 + (s0.sum(0, keepdim=True) - s0)) / n                              # :60
    fp = tp.sum(0, keepdim=True) - tp                                                 # :63-65
    fn = count - tp
    cost_siou = 1.0 - tp / (tp + fp + fn + 1e-6)                                      # :66-67
    with torch.no_grad():                                                             # reorder(), :42-52
        rows, cols = linear_sum_assignment((cost_ce + cost_siou).cpu().numpy())
        unmatched = np.array(sorted(set(range(ins_num)) - set(cols.tolist())), dtype=np.int64)
    rows_t = torch.as_tensor(rows, device=pred_ins.device)
    cols_t = torch.as_tensor(cols, device=pred_ins.device)
    valid_ce = cost_ce[rows_t, cols_t].mean()                                         # :27
    valid_siou = cost_siou[rows_t, cols_t].mean()                                     # :33
    if unmatched.size:                                                                # :29-32
        col_sums = tp.sum(0)
        invalid_ce = col_sums[torch.as_tensor(unmatched, device=pred_ins.device)].sum() / (n * unmatched.size)
    else:
        invalid_ce = torch.zeros((), device=pred_ins.device)
    return valid_ce + invalid_ce + valid_siou, valid_ce, invalid_ce, valid_siou
