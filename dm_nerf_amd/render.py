"""The render call with the reference's names and signatures (networks/render.py:6-96),
executed by the gfx950 kernels: fused field evaluation, LDS-staged compositing with a
wave-level transmittance scan, inverse-CDF resampling + merge."""
from __future__ import annotations

from typing import Dict

import torch

from . import ops
from .autograd import CompositeFn, FieldFn, needs_grad
from .dm_nerf import DM_NeRF, Embedder
from .helpers import _unit_linspace


def render_train(raw, z_vals, rays_d):
    """Alpha compositing ``(rgb_map, weights, depth_map, ins_map)`` (networks/render.py:6-28).
    ``ins_map`` has ``ins_num`` channels (the last, "air", logit channel is dropped, render.py:26)."""
    if torch.is_grad_enabled() and raw.requires_grad:
        return CompositeFn.apply(raw, z_vals, rays_d)          # HIP forward + HIP backward
    return ops.composite(raw, z_vals, rays_d, keep_last=False)


def _check_embedders(position_embedder, view_embedder):
    ok = (isinstance(position_embedder, Embedder) and isinstance(view_embedder, Embedder)
          and position_embedder.num_freqs == 10 and view_embedder.num_freqs == 4)
    if not ok:
        raise NotImplementedError(
            "the fused field kernel builds the 10-band position and 4-band view encodings itself "
            "(config.py:56-59); other embedders are not supported")


def dm_nerf(rays, position_embedder, view_embedder, model_coarse, model_fine, z_vals_coarse, args) -> Dict:
    """One chunk of rays: coarse field -> composite -> resample -> fine field -> composite.
    Same arguments and the same 10-key result dict as the reference (networks/render.py:31-96).

    ``args`` supplies ``perturb``, ``N_importance``, ``is_train`` and ``N_ins`` like the reference's
    namespace; optional ``args.t_rand`` / ``args.u`` replace the two ``torch.rand`` draws of the
    stochastic path (render.py:46, helpers.py:135) so that it can be checked against recorded draws.
    ``args.precision`` selects the field kernel's arithmetic: "bf16x6" (``ops.DEFAULT_PRECISION``: fp32 operands split
    exactly into three bf16 values, six bf16 MFMA products per fp32 product, fp32 accumulation -- fp32-grade), "fp32"
    (exact fp32 MFMA products), "fp16x3" or "bf16x3" (three 16-bit MFMAs per product on two-way splits: 22 / 16 operand
    bits).  With "fp16x3" the caller must look at ``ops.take_status`` afterwards (``render_image`` does): activations
    beyond the fp16 range are flagged, never silently wrong."""
    _check_embedders(position_embedder, view_embedder)
    if not (isinstance(model_coarse, DM_NeRF) and isinstance(model_fine, DM_NeRF)):
        raise TypeError("model_coarse / model_fine must be dm_nerf_amd.DM_NeRF modules")
    train = needs_grad(model_coarse, model_fine)
    rays_o, rays_d = rays
    precision = getattr(args, "precision", None) or ops.DEFAULT_PRECISION
    n_importance = int(args.N_importance)
    perturb = float(args.perturb)

    if perturb > 0.0:
        t_rand = getattr(args, "t_rand", None)
        if t_rand is None:
            t_rand = torch.rand(z_vals_coarse.shape, device=rays_d.device)
        z_vals_coarse = ops.jitter_depths(z_vals_coarse, t_rand)                       # render.py:40-47

    def field(model, z, sigma_only=False):
        if train:       # outputs carry grad to the parameters (SURVEY.md section 8b "Autograd")
            return FieldFn.apply(rays_o, rays_d, z, model, precision, *model.parameters())
        return ops.field_rays(rays_o, rays_d, z, model.blob(precision), model.ins_num, precision, sigma_only=sigma_only)

    # ``args.coarse_heads = False`` (set by render_image(image_only=True), inference in a split precision only): the
    # coarse network evaluates its trunk and density head only -- raw_coarse[..., 3] and with it the resampling weights
    # and everything fine are unchanged; rgb_coarse / ins_coarse come out as the composite of zeros.
    coarse_sigma_only = (not train) and precision != "fp32" and not getattr(args, "coarse_heads", True)
    raw_coarse = field(model_coarse, z_vals_coarse, coarse_sigma_only)
    rgb_coarse, weights_coarse, depth_coarse, ins_coarse = render_train(raw_coarse, z_vals_coarse, rays_d)
    weights_coarse = weights_coarse.detach()                                            # z_samples.detach(), render.py:68

    if n_importance > 0:
        if perturb == 0.0:
            u = _unit_linspace(n_importance, rays_d.device.index)                       # det=True, helpers.py:131-133
        else:
            u = getattr(args, "u", None)
            if u is None:
                u = torch.rand(rays_d.shape[0], n_importance, device=rays_d.device)
    else:
        u = None
    z_vals_fine = ops.resample_merge(z_vals_coarse, weights_coarse, n_importance, u)    # render.py:66-70

    raw_fine = field(model_fine, z_vals_fine)
    rgb_fine, _, depth_fine, ins_fine = render_train(raw_fine, z_vals_fine, rays_d)

    if args.is_train and getattr(args, "N_ins", None) is not None:                      # render.py:88-90
        ins_fine = ins_fine[-args.N_ins:]
        ins_coarse = ins_coarse[-args.N_ins:]
    return {"rgb_fine": rgb_fine, "ins_fine": ins_fine, "z_vals_fine": z_vals_fine, "raw_fine": raw_fine,
            "raw_coarse": raw_coarse, "rgb_coarse": rgb_coarse, "ins_coarse": ins_coarse,
            "z_vals_coarse": z_vals_coarse, "depth_fine": depth_fine, "depth_coarse": depth_coarse}
