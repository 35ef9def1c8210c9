"""Full-image rendering: the pose x chunk loop of the reference's ``render_test``
(networks/tester.py:54-85).  The metric / image-writing half of that function (SSIM, LPIPS,
Hungarian AP, PNG/JSON output, tester.py:87-168) is host-side evaluation and out of scope."""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch

import warnings
from types import SimpleNamespace

from . import ops
from .helpers import _unit_linspace, get_rays_k, z_val_sample
from .render import dm_nerf


def _resolved(args, image_only: bool):
    """``args`` with ``precision`` filled in (``ops.DEFAULT_PRECISION`` unless the caller said otherwise) and the coarse
    pass's heads switched off for ``image_only``."""
    precision = getattr(args, "precision", None) or ops.DEFAULT_PRECISION
    if image_only and precision == "fp32":
        raise ValueError("image_only (density-only coarse pass) is a mode of the bf16 / fp16 matrix-pipe kernels: set "
                         "args.precision to 'bf16x6', 'fp16x3' or 'bf16x3'")
    if getattr(args, "precision", None) != precision or image_only:
        args = SimpleNamespace(**{**vars(args), "precision": precision, "coarse_heads": not image_only})
    return args


def _render_rays(position_embedder, view_embedder, model_coarse, model_fine, rays_o, rays_d, args, chunk: int,
                 ray_range: Tuple[int, int], image_only: bool):
    """The chunk loop (tester.py:62-76) over flat rays ``[lo, hi)`` of ``rays_o`` / ``rays_d`` [H*W, 3]: rgb [hi-lo, 3] and
    instance map [hi-lo, ins_num], written chunk by chunk into preallocated buffers.  ``args.precision`` is resolved."""
    precision = args.precision
    lo, hi = ray_range
    n_total = hi - lo
    rgb = torch.empty(n_total, 3, dtype=torch.float32, device=rays_o.device)
    ins = torch.empty(n_total, model_fine.ins_num, dtype=torch.float32, device=rays_o.device)
    # Every arithmetic but the exact-fp32 one can run field evaluation, compositing and resampling fused (ops.render_fused:
    # two launches per chunk, `raw` never reaches memory, results bit-identical to dm_nerf()'s chain --
    # tests/test_gpu_parity.py).  Measured on one MI355X (65536-ray chunks), two-way splits (bf16x3): +5.6 % rays/s with
    # Replica's 98-channel rows, where the chain moves 100 KB of raw per ray through HBM with scattered row stores; -1 %
    # with DM-SR's 18 channels, where compositing is 0.6 % of a step and a lone wave per ray composites while its CU idles.
    # bf16x6 (twice the matrix work per point, so half the relative gain against the same idle waves): -0.9 % at 98
    # channels (650 k against 656 k rays/s).  So: on by default for the two-way splits with rows wider than the
    # lane-per-sample compositing form (ops.FUSED_MIN_CHANNELS); args.fused = True / False overrides.
    want = getattr(args, "fused", None)
    if want is None:
        want = precision in ops.X3_PRECISIONS and model_fine.ins_num + 5 > ops.FUSED_MIN_CHANNELS
    fused = (bool(want) and float(args.perturb) == 0.0
             and ops.fused_supported(int(args.N_samples), int(args.N_importance), model_fine.ins_num, precision))
    with torch.no_grad():
        for start in range(lo, hi, chunk):                                       # tester.py:62-71
            n = min(chunk, hi - start)
            z = z_val_sample(n, args.near, args.far, args.N_samples)
            if fused:
                out = ops.render_fused(rays_o[start:start + n], rays_d[start:start + n], z,
                                       model_coarse.blob(ops.save_blob_name(precision)), model_fine.blob(ops.save_blob_name(precision)), model_fine.ins_num,
                                       precision, int(args.N_importance), coarse_sigma_only=image_only,
                                       u=_unit_linspace(int(args.N_importance), rays_o.device.index))   # det=True
            else:
                batch = torch.stack([rays_o[start:start + n], rays_d[start:start + n]], dim=0)
                out = dm_nerf(batch, position_embedder, view_embedder, model_coarse, model_fine, z, args)
            rgb[start - lo:start - lo + n] = out["rgb_fine"]
            ins[start - lo:start - lo + n] = out["ins_fine"]
    return rgb, ins


def render_image(position_embedder, view_embedder, model_coarse, model_fine, c2w, hwk, args,
                 chunk: Optional[int] = None, crop_mask=None, ray_range: Optional[Tuple[int, int]] = None,
                 image_only: bool = False):
    """Render one pose.  ``chunk`` defaults to ``args.N_test`` (tester.py:62); the reference's
    O(chunks^2) ``torch.cat`` growth (tester.py:72-76) is replaced by writes into preallocated
    images.  ``ray_range=(a, b)`` renders only flat rays ``[a, b)`` (multi-GPU sharding of one
    image) and returns flat ``[b-a, .]`` tensors.  With rows wider than 32 channels the chunks of every arithmetic but the
    exact-fp32 one run through the fused field + compositing kernels (``ops.render_fused``).  ``image_only`` (the same
    arithmetics): the coarse network evaluates
    its density only -- the colour / instance heads of the coarse pass feed nothing this loop returns
    (tester.py:70-76 keeps ``rgb_fine`` and ``ins_fine``); the returned images are bit-identical."""
    H, W, K = hwk
    chunk = int(chunk or args.N_test)
    # fp32-grade arithmetic unless the caller says otherwise (ops.DEFAULT_PRECISION = "bf16x6": all 24 bits of every fp32
    # operand enter the products; same default as dm_nerf()).  args.precision = "fp32" selects the exact-fp32 MFMA kernel,
    # "fp16x3" | "bf16x3" the two-way split kernels (three 16-bit MFMAs per product: 1e-6 / 7e-6 on raw against the
    # reference, 1.7x the default's rays/s) -- narrower arithmetic is never chosen silently.
    args = _resolved(args, image_only)
    precision = args.precision
    if precision == "fp16x3":
        ops.take_status(torch.device("cuda", torch.cuda.current_device()))      # clear stale flags
    rays_o, rays_d = get_rays_k(H, W, K, c2w)                                   # tester.py:58
    rgb, ins = _render_rays(position_embedder, view_embedder, model_coarse, model_fine, rays_o.reshape(-1, 3),
                            rays_d.reshape(-1, 3), args, chunk, (0, H * W) if ray_range is None else ray_range, image_only)
    if precision == "fp16x3" and ops.take_status(rgb.device) & ops.STATUS_FP16_RANGE:
        # an activation left the fp16 range somewhere in this image: redo it with exact fp32 products
        warnings.warn("dm_nerf_amd: fp16x3 range exceeded, re-rendering this image with precision='fp32'")
        exact = SimpleNamespace(**{**vars(args), "precision": "fp32"})
        return render_image(position_embedder, view_embedder, model_coarse, model_fine, c2w, hwk, exact,
                            chunk=chunk, crop_mask=crop_mask, ray_range=ray_range)
    if ray_range is not None:
        return rgb, ins
    if crop_mask is not None:                                                    # tester.py:77-81
        keep = torch.as_tensor(crop_mask).reshape(-1).to(rgb.device) == 1
        rgb, ins = rgb[keep], ins[keep]
        return (rgb.reshape(args.crop_height, args.crop_width, 3),
                ins.reshape(args.crop_height, args.crop_width, ins.shape[-1]))
    return rgb.reshape(H, W, 3), ins.reshape(H, W, ins.shape[-1])


class GraphedImage:
    """The chunk loop of one image (networks/tester.py:62-76) captured as ONE hipGraph: every launch of every chunk -- at
    the reference's ``N_test = 4096`` that is 157 ``dm_nerf`` calls, ~1 900 kernels per 800x800 image -- replays with a single
    host call per image.  The device is not idle in the eager loop on an unloaded host (``bench.py``: ``chunk_4096`` runs at
    0.995 - 1.02 of the 65536-ray rate), so this buys nothing there; it is the form for a host that cannot keep ~2 000
    launches per image ahead of the device.

        renderer = GraphedImage(pos, view, coarse, fine, [H, W, K], args, chunk=4096)
        rgb, ins = renderer(c2w)          # [H, W, 3], [H, W, ins_num]: views of static buffers, overwritten by the next call

    Rays are generated eagerly per pose (the camera is a by-value kernel argument) into static buffers the graph reads.
    Inference only: the graph holds the weight blobs of the moment of capture -- call ``refresh()`` after the weights
    change.  Same kernels on the same values as ``render_image``: bit-identical images."""

    def __init__(self, position_embedder, view_embedder, model_coarse, model_fine, hwk, args, chunk: Optional[int] = None,
                 image_only: bool = False):
        self._nerf = (position_embedder, view_embedder, model_coarse, model_fine)
        self.hwk, self.args, self.chunk, self.image_only = hwk, _resolved(args, image_only), int(chunk or args.N_test), image_only
        if self.args.precision == "fp16x3":
            raise ValueError("GraphedImage: 'fp16x3' needs its range flag read back per image (render_image does); use "
                             "'bf16x6', 'fp32' or 'bf16x3'")
        if float(self.args.perturb) != 0.0:
            raise ValueError("GraphedImage renders the deterministic path (perturb = 0)")
        self.graph = None
        self.refresh()

    def refresh(self) -> None:
        H, W, K = self.hwk
        pos, view, coarse, fine = self._nerf
        dev = next(fine.parameters()).device
        eye = torch.eye(4)[:3].numpy()
        render_image(pos, view, coarse, fine, eye, self.hwk, self.args, chunk=self.chunk, image_only=self.image_only,
                     ray_range=(0, min(H * W, 2 * self.chunk)))              # warm-up: blobs, tables, kernels loaded
        self.rays_o = torch.empty(H * W, 3, dtype=torch.float32, device=dev)
        self.rays_d = torch.empty(H * W, 3, dtype=torch.float32, device=dev)
        self.rays_o.zero_(); self.rays_d.fill_(1.0)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            _render_rays(pos, view, coarse, fine, self.rays_o, self.rays_d, self.args, self.chunk, (0, H * W), self.image_only)
        torch.cuda.current_stream(dev).wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.rgb, self.ins = _render_rays(pos, view, coarse, fine, self.rays_o, self.rays_d, self.args, self.chunk,
                                              (0, H * W), self.image_only)

    def __call__(self, c2w):
        H, W, K = self.hwk
        rays_o, rays_d = get_rays_k(H, W, K, c2w)
        self.rays_o.copy_(rays_o.reshape(-1, 3))
        self.rays_d.copy_(rays_d.reshape(-1, 3))
        self.graph.replay()
        return self.rgb.reshape(H, W, 3), self.ins.reshape(H, W, self.ins.shape[-1])


def render_test(position_embedder, view_embedder, model_coarse, model_fine, render_poses, hwk, args, gt_imgs=None,
                gt_labels=None, ins_rgbs=None, savedir=None, matched_file=None, crop_mask=None, chunk=None):
    """Reference signature (networks/tester.py:18-19).  Renders every pose and returns
    ``(rgbs, inss, psnrs)``; PSNR against ``gt_imgs`` when given (tester.py:91).  The remaining
    arguments are accepted for call compatibility and ignored (evaluation / file output)."""
    rgbs: List[torch.Tensor] = []
    inss: List[torch.Tensor] = []
    psnrs: List[float] = []
    for i, c2w in enumerate(render_poses):
        rgb, ins = render_image(position_embedder, view_embedder, model_coarse, model_fine, c2w, hwk, args,
                                chunk=chunk, crop_mask=crop_mask)
        rgbs.append(rgb)
        inss.append(ins)
        if gt_imgs is not None:
            gt = torch.as_tensor(gt_imgs[i]).to(rgb.device)
            mse = torch.mean((rgb - gt) ** 2)
            psnrs.append(float(-10.0 * torch.log10(mse)))
    return rgbs, inss, psnrs
