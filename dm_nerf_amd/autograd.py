"""Autograd wiring of the HIP render path (training).

Forward of both functions is the hand-written HIP kernel.  Backward:

* ``CompositeFn``: hand-written HIP kernel (``dmn_composite_bwd``) with the reference's detach topology
  (the instance map is composited with ``weights.detach()``, networks/render.py:22-23).
* ``FieldFn``: hand-written HIP kernels (``csrc/field_bwd*.hip``): the forward writes a save record, a fused
  backward-data kernel chains dL/d pre-activation through the layers in registers, split-K MFMA kernels reduce
  the 30 weight/bias gradients over all sample points.  ``args.precision`` selects the arithmetic of forward AND
  backward under autograd: "bf16x6" (default; fp32-grade on the bf16 pipe) and "fp32" (exact fp32 MFMA) train with
  fp32-grade gradients, "bf16x3" / "fp16x3" (opt-in) with three bf16 MFMAs per product.
  Nothing here runs on the CPU; the library-GEMM cross-check of these kernels lives with the tests
  (``tests/library_backward.py``).

Only parameters receive gradients: rays and depths never need them (``z_samples.detach()``,
networks/render.py:68)."""
from __future__ import annotations

from typing import List

import torch

from . import ops
from ._lib import check, current_stream, load, ptr


class FieldFn(torch.autograd.Function):
    """raw = field(rays, z; params), hand-written HIP both ways.  Forward: the saving variant of the field kernel of
    ``precision`` -- "bf16x6" (the default: fp32 operands as exact three-way bf16 splits, six MFMA products per
    product), "fp32" (exact fp32 MFMA), or the two-way split "bf16x3" ("fp16x3" trains in bf16x3) -- which also writes
    the save record.  Backward = backward-data kernel (dL/d pre-activation of every layer, chained in registers) +
    split-K weight-gradient kernels + scatter into the flat gradient (``ops.field_backward``), in the arithmetic
    ``ops.backward_precision(precision)`` gives: fp32-grade for the fp32-grade forwards (the reference back-propagates
    in fp32, train_dmsr.py:62-64), bf16x3 only for the bf16x3 forward.  ``BACKWARD_OVERRIDE`` (module level; a pair
    (data, weights) of "fp32" | "bf16x6" | "bf16x3") replaces that choice for experiments and A/B timing."""

    @staticmethod
    def forward(ctx, rays_o, rays_d, z, model, precision, *params):
        # "fp16x3" trains in bf16x3: fp16 needs its range flag read back between forward and backward
        fwd = "bf16x6" if precision in ops.X6_PRECISIONS else ("fp32" if precision == "fp32" else "bf16x3")
        raw, save = ops.field_rays_save(rays_o, rays_d, z, model.blob(ops.save_blob_name(fwd)), model.ins_num, fwd)
        ctx.save, ctx.model = save, model
        ctx.backward_precisions = BACKWARD_OVERRIDE or (ops.backward_precision(fwd),) * 2
        ctx.meta = [(p.shape, p.numel(), p.requires_grad) for p in params]
        return raw

    @staticmethod
    def backward(ctx, g_raw):
        if ctx.save is None:
            raise RuntimeError("dm_nerf_amd: the field's save record is released by its first backward pass (it is "
                               "11.7 KB per sample point); a second backward through the same forward "
                               "(retain_graph=True) is not supported -- run the forward again")
        model = ctx.model
        n_params = sum(n for _, n, _ in ctx.meta)
        data_precision, wgrad_precision = ctx.backward_precisions
        flat = ops.field_backward(g_raw, ctx.save, model.blob(ops.BWD_BLOB[data_precision]), model.ins_num,
                                  model.wgrad_index(), n_params, wgrad_precision=wgrad_precision,
                                  data_precision=data_precision,
                                  out=model.lend_grad_buffer())  # a GradBucket's slice when one was adopted and is free
        ctx.save = None
        out: List = []
        at = 0
        for shape, n, wanted in ctx.meta:
            out.append(flat[at:at + n].view(shape) if wanted else None)
            at += n
        return (None, None, None, None, None, *out)


BACKWARD_OVERRIDE = None        # None: ops.backward_precision(forward); else (data_precision, wgrad_precision)


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
