"""The object manipulator's compute (networks/manipulator.py:18-205): field evaluation and compositing outside
``dm_nerf`` (``manipulator_nerf`` / ``manipulator_render``), the label surgery between the original and the
transformed scene (``exchanger``) and the two-pass re-render that strings them together (``manipulator``).  The demo /
evaluation drivers around them (image and json writers, metrics) are out of scope.
Pinned by G12 (``tests/golden/make_golden.py:g12_manipulator``): the reference's own four function definitions,
compiled from its source text, run on the seeded cases; ``exchanger`` is held to bit equality, the re-render to the
path tolerances."""
from __future__ import annotations

import torch

from . import ops
from ._lib import check, current_stream, load, ptr
from .helpers import sample_pdf
from .render import _check_embedders


def manipulator_render(raw, z_vals, rays_d):
    """Compositing variant that keeps all ``ins_num + 1`` instance channels
    (networks/manipulator.py:86-105).  Returns ``(rgb_map, weights, depth_map, ins_map)``."""
    if torch.is_grad_enabled() and raw.requires_grad:
        raise NotImplementedError("manipulator_render is an inference path (it composites with undetached weights); "
                                  "use torch.no_grad()")
    return ops.composite(raw, z_vals, rays_d, keep_last=True)


def manipulator_nerf(rays, position_embedder, view_embedder, model, N_samples=None, near=None, far=None, z_vals=None):
    """Field evaluation on rays at given or lattice depths (networks/manipulator.py:108-134).
    Returns ``(raw, z_vals)``."""
    _check_embedders(position_embedder, view_embedder)
    model._refuse_grad()
    rays_o, rays_d = rays
    n = rays_d.shape[0]
    if z_vals is None:
        # z = near*(1-t) + far*t (manipulator.py:117-120): same host arithmetic as the reference
        t = torch.linspace(0.0, 1.0, steps=int(N_samples))
        row = (near * torch.ones(1, 1)) * (1.0 - t) + (far * torch.ones(1, 1)) * t
        z_vals = row.to(rays_d.device).expand([n, int(N_samples)])
    raw = ops.field_rays(rays_o, rays_d, z_vals, model.blob(), model.ins_num)
    return raw, z_vals


def _labels(x: torch.Tensor, rows: int, row_stride: int, k: int, offset: int = 0) -> torch.Tensor:
    out = torch.empty(rows, dtype=torch.int32, device=x.device)
    check(load().dmn_sigmoid_argmax(x.data_ptr() + 4 * offset, rows, row_stride, k, ptr(out), current_stream()),
          "dmn_sigmoid_argmax")
    return out


def exchanger(ori_raw, tar_raws, ori_raw_pred, tar_raw_preds, move_labels):
    """``exchanger(ori_raw, tar_raws, ori_raw_pred, tar_raw_preds, move_labels)`` (networks/manipulator.py:18-83):
    for every moved object, points of the original scene that carry its label are emptied, points where the
    transformed scene carries it take the transformed scene's raw row, and points behind which the ray ends on the
    object are filled from the transformed scene.  ``ori_raw`` is edited in place, like the reference's.  Returns
    ``(ori_raw, tar_raws, ori_pred_label, tar_pred_label)`` (labels int64 ``[N, S]``; the last object's target labels)."""
    ops._dev(ori_raw, "ori_raw")
    if not ori_raw.is_contiguous():
        raise ValueError("ori_raw must be contiguous (it is edited in place)")
    n, s, c = ori_raw.shape
    ori_pred = ops._dev(ori_raw_pred, "ori_raw_pred").contiguous()
    ori_labels = _labels(ori_raw, n * s, c, c - 4, offset=4)                                   # :19-21
    ori_ray = _labels(ori_pred, n, ori_pred.shape[-1], ori_pred.shape[-1] - 1)                 # :23-26
    tar_labels = torch.zeros(n * s, dtype=torch.int32, device=ori_raw.device)
    for tar_raw, tar_pred, move in zip(tar_raws, tar_raw_preds, move_labels):                  # :28
        tar_raw = ops._dev(tar_raw, "tar_raw").contiguous()
        if tuple(tar_raw.shape) != (n, s, c):
            raise ValueError("tar_raw must have the shape of ori_raw")
        tar_pred = ops._dev(tar_pred, "tar_raw_pred").contiguous()
        tar_ray = _labels(tar_pred, n, tar_pred.shape[-1], tar_pred.shape[-1] - 1)             # :50-53
        check(load().dmn_exchange(ptr(ori_raw), ptr(tar_raw), ptr(ori_labels), ptr(ori_ray), ptr(tar_ray),
                                  ptr(tar_labels), n, s, c, int(move), current_stream()), "dmn_exchange")
    return ori_raw, tar_raws, ori_labels.reshape(n, s).long(), tar_labels.reshape(n, s).long()


def manipulator(position_embedder, view_embedder, model_coarse, model_fine, ori_rays, f_tar_rays, args):
    """Two-pass re-render of a scene with moved objects (networks/manipulator.py:136-205): coarse fields of the
    original rays and of every object's inversely transformed rays, their composited labels at 64 + 128 depths, the
    exchange, then the same at the union of all resampled depths with the fine field.  ``args``: ``N_samples``,
    ``N_importance``, ``near``, ``far``, ``target_label`` (one label per entry of ``f_tar_rays``); optional
    ``args.u_draws``: the uniform draws of the ``2 + len(f_tar_rays)`` ``sample_pdf`` calls in call order (the reference
    draws them with the global ``torch.rand``, helpers.py:135), so that a run can be checked against recorded draws.
    Returns ``(final_rgb, final_ins, tar_rgb, tar_ins_accum)``."""
    n_samples, n_importance, near, far = args.N_samples, args.N_importance, args.near, args.far
    draws = list(getattr(args, "u_draws", None) or [])

    def resampled(z, weights):
        mid = 0.5 * (z[..., 1:] + z[..., :-1])
        u = draws.pop(0) if draws else None
        return sample_pdf(mid, weights[..., 1:-1], n_importance, u=u)                          # det=False, :146,168

    def labels_at(rays, z, z_new):
        z_full = ops.merge_depths(z, z_new)                                                    # sort(cat), :148
        raw_full, _ = manipulator_nerf(rays, position_embedder, view_embedder, model_fine, z_vals=z_full)
        return manipulator_render(raw_full, z_full, rays[1])[3]

    with torch.no_grad():
        ori_raw, ori_z = manipulator_nerf(ori_rays, position_embedder, view_embedder, model_coarse, n_samples, near, far)
        ori_raw = ori_raw.contiguous()
        ori_z = ori_z.contiguous()
        _, ori_w, _, _ = manipulator_render(ori_raw, ori_z, ori_rays[1])
        ori_ins_accum = labels_at(ori_rays, ori_z, resampled(ori_z, ori_w))                    # :142-150
        tar_raws, tar_zs, tar_samples, tar_ins_accums = [], [], [], []
        tar_rgb = tar_ins_accum = None
        for tar_rays in f_tar_rays:                                                            # :153-176
            tar_raw, tar_z = manipulator_nerf(tar_rays, position_embedder, view_embedder, model_coarse, n_samples,
                                              near, far)
            tar_rgb, tar_w, _, _ = manipulator_render(tar_raw, tar_z, tar_rays[1])
            z_new = resampled(tar_z.contiguous(), tar_w)
            tar_ins_accum = labels_at(tar_rays, tar_z, z_new)
            tar_raws.append(tar_raw); tar_zs.append(tar_z); tar_samples.append(z_new); tar_ins_accums.append(tar_ins_accum)
        ori_raw, _, _, _ = exchanger(ori_raw, tar_raws, ori_ins_accum, tar_ins_accums, args.target_label)   # :179
        # step 2 (:181-203): weights of the exchanged scene, all resampled depths merged, fine fields, exchange again
        _, ori_w, _, _ = manipulator_render(ori_raw, ori_z, ori_rays[1])
        ori_new = resampled(ori_z, ori_w)
        extra = torch.cat([ori_new] + tar_samples, -1)                                         # :190-191
        ori_z_all = ops.merge_depths(ori_z, extra)
        ori_raw, _ = manipulator_nerf(ori_rays, position_embedder, view_embedder, model_fine, z_vals=ori_z_all)
        for idx, tar_rays in enumerate(f_tar_rays):
            tar_z_all = ops.merge_depths(tar_zs[idx], extra)                                   # :197
            tar_raws[idx], _ = manipulator_nerf(tar_rays, position_embedder, view_embedder, model_fine, z_vals=tar_z_all)
        ori_raw, _, _, _ = exchanger(ori_raw.contiguous(), tar_raws, ori_ins_accum, tar_ins_accums, args.target_label)
        final_rgb, _, _, final_ins = manipulator_render(ori_raw, ori_z_all, ori_rays[1])
    return final_rgb, final_ins, tar_rgb, tar_ins_accum
