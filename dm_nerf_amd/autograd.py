"""Autograd wiring of the HIP render path (training).

Forward of both functions is the hand-written HIP kernel.  Backward:

* ``CompositeFn``: hand-written HIP kernel (``dmn_composite_bwd``) with the reference's detach topology
  (the instance map is composited with ``weights.detach()``, networks/render.py:22-23).
* ``FieldFn``: hand-written HIP kernels (``csrc/field_bwd.hip``): the forward writes a save record, a fused
  backward-data kernel chains dL/d pre-activation through the layers in registers, a split-K MFMA kernel reduces
  the 30 weight/bias gradients over all sample points.  ``args.precision`` selects the forward under autograd:
  exact fp32, or three bf16 MFMAs per product for "bf16x3" / "fp16x3".
  ``FieldFnLibrary`` (library GEMMs on recomputed activations, ``torch_field_on_rays``) is the cross-check.
  Nothing here runs on the CPU.

Only parameters receive gradients: rays and depths never need them (``z_samples.detach()``,
networks/render.py:68)."""
from __future__ import annotations

from typing import Dict, List

import torch
import torch.nn.functional as F

from . import ops
from ._lib import check, current_stream, load, ptr

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


class FieldFn(torch.autograd.Function):
    """raw = field(rays, z; params), hand-written HIP both ways: the exact-fp32 forward kernel also writes its save
    record; backward = backward-data kernel (dL/d pre-activation of every layer, chained in registers) + split-K
    weight-gradient kernel + scatter into the flat gradient (``ops.field_backward``)."""

    @staticmethod
    def forward(ctx, rays_o, rays_d, z, model, precision, *params):
        # "fp16x3" trains in bf16x3: fp16 needs its range flag read back between forward and backward
        fwd = "fp32" if precision == "fp32" else "bf16x3"
        raw, save = ops.field_rays_save(rays_o, rays_d, z, model.blob(fwd), model.ins_num, fwd)
        ctx.save, ctx.model = save, model
        ctx.meta = [(p.shape, p.numel(), p.requires_grad) for p in params]
        return raw

    @staticmethod
    def backward(ctx, g_raw):
        model = ctx.model
        n_params = sum(n for _, n, _ in ctx.meta)
        blob = model.blob("bwd" if DATA_PRECISION == "fp32" else "bwd_x3")
        flat = ops.field_backward(g_raw, ctx.save, blob, model.ins_num, model.wgrad_index(), n_params,
                                  wgrad_precision=WGRAD_PRECISION, data_precision=DATA_PRECISION)
        ctx.save = None
        out: List = []
        at = 0
        for shape, n, wanted in ctx.meta:
            out.append(flat[at:at + n].view(shape) if wanted else None)
            at += n
        return (None, None, None, None, None, *out)


class FieldFnLibrary(torch.autograd.Function):
    """Same function with the round-1 backward: parameter gradients through library GEMMs (rocBLAS via torch) on
    activations recomputed from the saved rays and depths.  Kept as the on-device cross-check of ``FieldFn``
    (tests) and for A/B timing (``FIELD_BACKWARD = "library"``)."""

    @staticmethod
    def forward(ctx, rays_o, rays_d, z, model, precision, *params):
        raw = ops.field_rays(rays_o, rays_d, z, model.blob(precision), model.ins_num, precision)
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


FIELD_BACKWARD = "hip"          # "library": FieldFnLibrary
DATA_PRECISION = "bf16x3"       # backward-data chain: "bf16x3" or "fp32" (exact MFMA); masks come from the save record
WGRAD_PRECISION = "bf16x3"      # weight-gradient GEMMs: "bf16x3" (three bf16 MFMAs per product) or "fp32" (exact MFMA)


def field_fn():
    return FieldFn if FIELD_BACKWARD == "hip" else FieldFnLibrary


class CompositeFn(torch.autograd.Function):
    """(rgb_map, weights, depth_map, ins_map) = render_train(raw, z, rays_d); HIP forward and backward."""

    @staticmethod
    def forward(ctx, raw, z, rays_d):
        ctx.set_materialize_grads(False)          # unused outputs arrive as None -> NULL in the C call
        rgb, weights, depth, ins = ops.composite(raw, z, rays_d, keep_last=False)
        ctx.save_for_backward(raw, z.contiguous(), rays_d, ins)
        return rgb, weights, depth, ins

    @staticmethod
    def backward(ctx, g_rgb, g_w, g_depth, g_ins):
        raw, z, rays_d, ins = ctx.saved_tensors
        n, s, c = raw.shape
        g_raw = torch.empty_like(raw)
        cont = lambda t: None if t is None else t.contiguous()
        g_rgb, g_w, g_depth, g_ins = cont(g_rgb), cont(g_w), cont(g_depth), cont(g_ins)
        check(load().dmn_composite_bwd(ptr(raw), ptr(z), s, ptr(rays_d.contiguous()), ptr(ins), ptr(g_rgb), ptr(g_depth),
                                       ptr(g_ins), ptr(g_w), n, s, c, 0, ptr(g_raw), current_stream()),
              "dmn_composite_bwd")
        return g_raw, None, None


def needs_grad(*models) -> bool:
    return torch.is_grad_enabled() and any(p.requires_grad for m in models for p in m.parameters())
