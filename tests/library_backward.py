"""On-device cross-check of the hand-written field backward (test infrastructure, not product): the same function
with parameter gradients through library GEMMs (rocBLAS via torch autograd) on activations recomputed from the rays
and depths, plus the torch-op restatement of the field it differentiates (networks/render.py:37-38,49-61,
networks/dm_nerf.py:80-106)."""
from __future__ import annotations

from typing import Dict, List

import torch
import torch.nn.functional as F

from dm_nerf_amd import ops

POS_FREQS, DIR_FREQS, POS_CH = 10, 4, 63


def _encode(x: torch.Tensor, n_freqs: int) -> torch.Tensor:
    """networks/dm_nerf.py:17-38 with torch ops on the device (backward recompute only)."""
    parts = [x]
    for k in range(n_freqs):
        parts.append(torch.sin(x * float(2 ** k)))
        parts.append(torch.cos(x * float(2 ** k)))
    return torch.cat(parts, -1)


def torch_field_on_rays(p: Dict[str, torch.Tensor], rays_o, rays_d, z) -> torch.Tensor:
    """The field network on ``o + d*z`` with differentiable torch ops (networks/render.py:37-38,49-61,
    networks/dm_nerf.py:80-106): used to rebuild the activations for the parameter gradients."""
    view = rays_d / torch.norm(rays_d, dim=-1, keepdim=True)
    pts = rays_o[:, None, :] + rays_d[:, None, :] * z[:, :, None]
    pos = _encode(pts.reshape(-1, 3), POS_FREQS)
    vdir = _encode(view[:, None, :].expand(pts.shape).reshape(-1, 3), DIR_FREQS)
    h = pos
    for i in range(8):
        h = F.relu(F.linear(h, p[f"mlps.{i}.weight"], p[f"mlps.{i}.bias"]))
        if i == 4:
            h = torch.cat([h, pos], -1)
    feat = F.linear(h, p["rgb_feature_linear.weight"], p["rgb_feature_linear.bias"])
    feat = F.relu(F.linear(torch.cat([feat, vdir], -1), p["rgb_feature_linears.0.weight"],
                           p["rgb_feature_linears.0.bias"]))
    code = F.linear(h.detach(), p["ins_feature_linear.weight"], p["ins_feature_linear.bias"])   # dm_nerf.py:95
    code = F.relu(F.linear(code, p["ins_feature_linears.0.weight"], p["ins_feature_linears.0.bias"]))
    sigma = F.linear(h, p["density_linear.weight"], p["density_linear.bias"])
    rgb = F.linear(feat, p["rgb_linear.weight"], p["rgb_linear.bias"])
    ins = F.linear(code, p["ins_linear.weight"], p["ins_linear.bias"])
    return torch.cat([rgb, sigma, ins], -1).reshape(z.shape[0], z.shape[1], -1)


class FieldFnLibrary(torch.autograd.Function):
    """Same function with the round-1 backward: parameter gradients through library GEMMs (rocBLAS via torch) on
    activations recomputed from the saved rays and depths.  Kept as the on-device cross-check of ``FieldFn``
    (tests) and for A/B timing (``FIELD_BACKWARD = "library"``)."""

    @staticmethod
    def forward(ctx, rays_o, rays_d, z, model, precision, *params):
        kernel = ops.save_blob_name(precision)      # the kernel family FieldFn's saving forward runs (bf16x6: 32x32x16)
        raw = ops.field_rays(rays_o, rays_d, z, model.blob(kernel), model.ins_num, kernel)
        ctx.save_for_backward(rays_o, rays_d, z.contiguous())
        ctx.names = [n for n, _ in model.named_parameters()]
        ctx.params = params
        return raw

    @staticmethod
    def backward(ctx, g_raw):
        rays_o, rays_d, z = ctx.saved_tensors
        leaves = [p.detach().requires_grad_(p.requires_grad) for p in ctx.params]
        with torch.enable_grad():
            raw = torch_field_on_rays(dict(zip(ctx.names, leaves)), rays_o, rays_d, z)
        wanted = [t for t in leaves if t.requires_grad]
        grads = torch.autograd.grad(raw, wanted, g_raw.contiguous(), allow_unused=True)
        it = iter(grads)
        out: List = [next(it) if t.requires_grad else None for t in leaves]
        return (None, None, None, None, None, *out)
