"""CPU oracle for the DM-NeRF volumetric-render hot path.

TEST INFRASTRUCTURE ONLY.  This module is a CPU restatement (PyTorch CPU ops, fp32) of
the reference algorithm.  It is imported only by ``tests/``, ``__graft_entry__.smoke()``
and the ``cpu_baseline`` leg of ``bench.py`` as the *checker*; the product package
``dm_nerf_amd`` never imports it and has no CPU fallback.

Parity pin: every function here is checked in ``tests/test_oracle.py`` against golden
vectors produced by importing the reference itself (``tests/golden/make_golden.py``,
run in the build container where ``/root/reference`` is mounted; the vectors are
committed under ``tests/golden/*.npz``).  The reference ships no tests or fixtures of
its own.

Each function cites the reference lines it restates (paths relative to the reference
root).  The op *sequence* follows the reference (per-layer ``addmm``, ``cat``-based
encoding, ``cumprod`` transmittance, ``searchsorted`` sampler) so that timing this
module is a fair stand-in for the reference's PyTorch-CPU path.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Params = Dict[str, torch.Tensor]

NET_DEPTH = 8          # config.py:23
NET_WIDTH = 256        # config.py:25
SKIP_AFTER = 4         # config.py:153,157 -> skips=[4]
POS_FREQS = 10         # config.py:56
DIR_FREQS = 4          # config.py:58
POS_CH = 3 + 6 * POS_FREQS   # 63
DIR_CH = 3 + 6 * DIR_FREQS   # 27


# --------------------------------------------------------------------------------------
# parameters
# --------------------------------------------------------------------------------------
# Seeded synthetic weights (data, not algorithm) come from the product-side generator so that both sides of a parity
# test build identical tensors from a seed: the oracle may import the product, never the other way round.
from dm_nerf_amd.synthetic import (dmsr_intrinsics, make_params, param_specs, pose_spherical,   # noqa: E402,F401
                                   study_pose)


# --------------------------------------------------------------------------------------
# rays and depths
# --------------------------------------------------------------------------------------
def rays_from_camera(H: int, W: int, K, c2w: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Per-pixel ray origins and (un-normalised) directions, ``[H, W, 3]`` each.

    Restates ``get_rays_k`` (networks/helpers.py:50-61): pixel centres at integer
    coordinates, camera-frame direction ``((w-K02)/K00, (h-K12)/K11, K22)``, rotated by
    ``c2w[:3,:3]``; origin ``c2w[:3,-1]`` broadcast.
    """
    K = np.asarray(K)
    cols = torch.arange(W, dtype=torch.float32).view(1, W).expand(H, W)
    rows = torch.arange(H, dtype=torch.float32).view(H, 1).expand(H, W)
    cam = torch.stack([(cols - K[0, 2]) / K[0, 0],
                       (rows - K[1, 2]) / K[1, 1],
                       K[2, 2] * torch.ones_like(cols)], dim=-1)
    rot = c2w[:3, :3]
    rays_d = (cam.unsqueeze(-2) * rot).sum(-1)
    rays_o = c2w[:3, -1].expand(rays_d.shape)
    return rays_o, rays_d


def coarse_depths(n_rays: int, near: float, far: float, n_samples: int) -> torch.Tensor:
    """``z = near + linspace(0,1,n)*(far-near)`` expanded to ``[n_rays, n]``
    (``z_val_sample``, networks/helpers.py:114-119)."""
    lo = near * torch.ones(n_rays, 1)
    hi = far * torch.ones(n_rays, 1)
    t = torch.linspace(0.0, 1.0, steps=n_samples)
    return (lo + t * (hi - lo)).expand(n_rays, n_samples)


def jitter_depths(z: torch.Tensor, t_rand: torch.Tensor) -> torch.Tensor:
    """Stratified jitter inside the coarse bins (networks/render.py:40-47)."""
    mid = 0.5 * (z[..., 1:] + z[..., :-1])
    upper = torch.cat([mid, z[..., -1:]], -1)
    lower = torch.cat([z[..., :1], mid], -1)
    return lower + (upper - lower) * t_rand


# --------------------------------------------------------------------------------------
# field
# --------------------------------------------------------------------------------------
def encode(x: torch.Tensor, n_freqs: int) -> torch.Tensor:
    """``[x, sin(2^0 x), cos(2^0 x), ..., sin(2^(L-1) x), cos(2^(L-1) x)]``
    (``Embedder.embed``, networks/dm_nerf.py:13-38; log-sampled bands ``:25``)."""
    bands = 2.0 ** torch.linspace(0.0, n_freqs - 1, steps=n_freqs)
    parts = [x]
    for f in bands:
        parts.append(torch.sin(x * f))
        parts.append(torch.cos(x * f))
    return torch.cat(parts, -1)


def field(params: Params, embedded: torch.Tensor) -> torch.Tensor:
    """Field network ``[M, 63+27] -> [M, 3+1+ins_num+1]`` (``DM_NeRF.forward``,
    networks/dm_nerf.py:80-106).  Raw outputs, no activations on the heads."""
    pos, view = embedded[..., :POS_CH], embedded[..., POS_CH:POS_CH + DIR_CH]
    h = pos
    for i in range(NET_DEPTH):
        h = F.relu(F.linear(h, params[f"mlps.{i}.weight"], params[f"mlps.{i}.bias"]))
        if i == SKIP_AFTER:
            h = torch.cat([h, pos], -1)                                   # dm_nerf.py:86-87
    feat = F.linear(h, params["rgb_feature_linear.weight"], params["rgb_feature_linear.bias"])
    feat = torch.cat([feat, view], -1)                                    # dm_nerf.py:90
    feat = F.relu(F.linear(feat, params["rgb_feature_linears.0.weight"], params["rgb_feature_linears.0.bias"]))
    code = F.linear(h.detach(), params["ins_feature_linear.weight"], params["ins_feature_linear.bias"])  # :95-96
    code = F.relu(F.linear(code, params["ins_feature_linears.0.weight"], params["ins_feature_linears.0.bias"]))
    sigma = F.linear(h, params["density_linear.weight"], params["density_linear.bias"])
    rgb = F.linear(feat, params["rgb_linear.weight"], params["rgb_linear.bias"])
    ins = F.linear(code, params["ins_linear.weight"], params["ins_linear.bias"])
    return torch.cat([rgb, sigma, ins], -1)


def field_on_rays(params: Params, rays_o: torch.Tensor, rays_d: torch.Tensor, z: torch.Tensor) -> torch.Tensor:
    """Points ``o + d*z`` -> encodings -> field, ``[N, S, C]`` (networks/render.py:37-38,49-61)."""
    view = rays_d / torch.norm(rays_d, dim=-1, keepdim=True)
    pts = rays_o[..., None, :] + rays_d[..., None, :] * z[..., :, None]
    flat = pts.reshape(-1, 3)
    emb = torch.cat([encode(flat, POS_FREQS),
                     encode(view[:, None].expand(pts.shape).reshape(-1, 3), DIR_FREQS)], -1)
    raw = field(params, emb)
    return raw.reshape(*pts.shape[:-1], raw.shape[-1])


# --------------------------------------------------------------------------------------
# compositing and resampling
# --------------------------------------------------------------------------------------
def composite(raw: torch.Tensor, z: torch.Tensor, rays_d: torch.Tensor, manipulator: bool = False):
    """Alpha compositing.  Returns ``(rgb[N,3], weights[N,S], depth[N], ins[N,*])``.

    ``manipulator=False`` restates ``render_train`` (networks/render.py:6-28): instance map is
    ``sigmoid(sum_i w_i.detach() * logits_i)`` with the last ("air") channel dropped.
    ``manipulator=True`` restates ``manipulator_render`` (networks/manipulator.py:86-105):
    no detach and all ``ins_num+1`` channels kept.
    """
    delta = z[..., 1:] - z[..., :-1]
    delta = torch.cat([delta, torch.full_like(delta[..., :1], 1e10)], -1)
    delta = delta * torch.norm(rays_d[..., None, :], dim=-1)
    colour = torch.sigmoid(raw[..., :3])
    alpha = 1.0 - torch.exp(-F.relu(raw[..., 3]) * delta)
    trans = torch.cumprod(torch.cat([torch.ones_like(alpha[:, :1]), 1.0 - alpha + 1e-10], -1), -1)[:, :-1]
    weights = alpha * trans
    rgb_map = torch.sum(weights[..., None] * colour, -2)
    depth_map = torch.sum(weights * z, -1)
    w_ins = weights if manipulator else weights.detach()
    ins_map = torch.sigmoid(torch.sum(w_ins[..., None] * raw[..., 4:], -2))
    if not manipulator:
        ins_map = ins_map[..., :-1]
    return rgb_map, weights, depth_map, ins_map


def inverse_cdf(bins: torch.Tensor, weights: torch.Tensor, n_new: int, det: bool,
                u: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Hierarchical sampler (``sample_pdf``, networks/helpers.py:123-155).

    ``u`` overrides the uniform draws when ``det`` is False (the reference calls the global
    ``torch.rand``; parity tests record its output and pass it here)."""
    w = weights + 1e-5
    pdf = w / torch.sum(w, -1, keepdim=True)
    cdf = torch.cumsum(pdf, -1)
    cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], -1)
    if det:
        u = torch.linspace(0.0, 1.0, steps=n_new).expand(*cdf.shape[:-1], n_new)
    elif u is None:
        u = torch.rand(*cdf.shape[:-1], n_new)
    u = u.contiguous()
    idx = torch.searchsorted(cdf, u, right=True)
    lo = torch.clamp(idx - 1, min=0)
    hi = torch.clamp(idx, max=cdf.shape[-1] - 1)
    cdf_lo, cdf_hi = torch.gather(cdf, -1, lo), torch.gather(cdf, -1, hi)
    bin_lo, bin_hi = torch.gather(bins, -1, lo), torch.gather(bins, -1, hi)
    den = cdf_hi - cdf_lo
    den = torch.where(den < 1e-5, torch.ones_like(den), den)
    t = (u - cdf_lo) / den
    return bin_lo + t * (bin_hi - bin_lo)


def merge_depths(z_coarse: torch.Tensor, z_new: torch.Tensor) -> torch.Tensor:
    """Ascending merge of coarse and resampled depths (networks/render.py:70)."""
    return torch.sort(torch.cat([z_coarse, z_new], -1), -1).values


# --------------------------------------------------------------------------------------
# the render call
# --------------------------------------------------------------------------------------
def render_rays(rays: torch.Tensor, coarse: Params, fine: Params, z_coarse: torch.Tensor,
                n_importance: int = 128, perturb: float = 0.0,
                t_rand: Optional[torch.Tensor] = None, u: Optional[torch.Tensor] = None,
                n_ins: Optional[int] = None) -> Dict[str, torch.Tensor]:
    """One chunk of rays through coarse field -> composite -> resample -> fine field ->
    composite.  Restates ``dm_nerf`` (networks/render.py:31-96); returns its 10-key dict plus
    ``weights_coarse`` / ``weights_fine`` (extra keys, used by stage-wise parity tests)."""
    rays_o, rays_d = rays[0], rays[1]
    if perturb > 0.0:
        if t_rand is None:
            t_rand = torch.rand(z_coarse.shape)
        z_coarse = jitter_depths(z_coarse, t_rand)
    raw_c = field_on_rays(coarse, rays_o, rays_d, z_coarse)
    rgb_c, w_c, depth_c, ins_c = composite(raw_c, z_coarse, rays_d)
    mid = 0.5 * (z_coarse[..., 1:] + z_coarse[..., :-1])
    z_new = inverse_cdf(mid, w_c[..., 1:-1], n_importance, det=(perturb == 0.0), u=u).detach()
    z_fine = merge_depths(z_coarse, z_new)
    raw_f = field_on_rays(fine, rays_o, rays_d, z_fine)
    rgb_f, w_f, depth_f, ins_f = composite(raw_f, z_fine, rays_d)
    if n_ins is not None:                                                  # render.py:88-90
        ins_f, ins_c = ins_f[-n_ins:], ins_c[-n_ins:]
    return {"rgb_fine": rgb_f, "ins_fine": ins_f, "z_vals_fine": z_fine, "raw_fine": raw_f,
            "raw_coarse": raw_c, "rgb_coarse": rgb_c, "ins_coarse": ins_c, "z_vals_coarse": z_coarse,
            "depth_fine": depth_f, "depth_coarse": depth_c,
            "weights_coarse": w_c, "weights_fine": w_f}


def render_image(H: int, W: int, K, c2w: torch.Tensor, coarse: Params, fine: Params,
                 near: float, far: float, n_samples: int = 64, n_importance: int = 128,
                 chunk: int = 4096, max_rays: Optional[int] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Full-image render in ``chunk``-ray pieces with a ragged tail; returns
    ``(rgb[H,W,3], ins[H,W,ins_num])``.  Restates the chunk loop of ``render_test``
    (networks/tester.py:57-85, no crop mask).  ``max_rays`` truncates the loop (used by the
    bounded CPU-baseline timing); the outputs are then flat ``[max_rays, .]``."""
    z = coarse_depths(chunk, near, far, n_samples)
    rays_o, rays_d = rays_from_camera(H, W, K, c2w)
    rays_o = rays_o.reshape(-1, 3).float()
    rays_d = rays_d.reshape(-1, 3).float()
    total = H * W if max_rays is None else min(max_rays, H * W)
    rgbs, inss = [], []
    with torch.no_grad():
        for start in range(0, total, chunk):
            n = min(chunk, total - start)
            if n != z.shape[0]:
                z = coarse_depths(n, near, far, n_samples)
            out = render_rays(torch.stack([rays_o[start:start + n], rays_d[start:start + n]], 0),
                              coarse, fine, z, n_importance=n_importance, perturb=0.0)
            rgbs.append(out["rgb_fine"])
            inss.append(out["ins_fine"])
    rgb, ins = torch.cat(rgbs, 0), torch.cat(inss, 0)
    if max_rays is None:
        rgb, ins = rgb.reshape(H, W, 3), ins.reshape(H, W, ins.shape[-1])
    return rgb, ins


# --------------------------------------------------------------------------------------
# synthetic cameras (conventions of datasets/loader_dmsr.py:29-34,171-172)
# --------------------------------------------------------------------------------------
# --------------------------------------------------------------------------------------
# training loss head on the path's raw output (row N2)
# --------------------------------------------------------------------------------------
def emptiness_penalizer(raw: torch.Tensor, z: torch.Tensor, depth: torch.Tensor, rays_d: torch.Tensor,
                        tolerance: float, deta_w: float) -> torch.Tensor:
    """Emptiness loss on the per-sample instance logits (networks/penalizer.py:5-55; ``ins_penalizer`` :58-62
    passes ``depth[..., None].detach()``).  ``raw [N,S,4+ins_num+1]``, ``z [N,S]``, ``depth [N]`` -> scalar.

    Samples more than ``tolerance`` in front of the rendered depth should be "air" (last channel on, all others
    off, weighted by 1 - Gaussian of the distance to the depth); samples within the tolerance band should not be
    air (weighted by the Gaussian); samples behind are free."""
    depth = depth.detach()[..., None]
    norm = torch.norm(rays_d[..., None, :], dim=-1)                                   # :13
    dist_before, dist_after = (depth - tolerance) * norm, (depth + tolerance) * norm  # :14-17
    p = z * norm                                                                      # :19
    delta = depth * norm - p                                                          # :18,22
    width = torch.tensor([deta_w])
    gauss = torch.exp(-(delta ** 2) / (2 * (width ** 2))) / (torch.tensor([0.4]) * torch.sqrt(torch.tensor([2 * np.pi]))) + 1e-8
    before = (p < dist_before).float()                                                # :27
    after = (p > dist_after).float()
    middle = 1 - (after + before)                                                     # :29
    pred = torch.sigmoid(raw[..., 4:])                                                # :32-33
    target = torch.zeros_like(pred)
    target[..., -1] = 1                                                               # :37-38
    bce = -target * torch.log(pred + 1e-8) - (1 - target) * torch.log(1 - pred + 1e-8)
    loss_before = torch.sum(bce * ((1 - gauss) * before)[..., None]) / (
        pred.shape[-1] * torch.maximum(torch.sum(before), torch.tensor([1e-8])))      # :40-43
    air = pred[..., -1]
    loss_middle = torch.sum(-torch.log(1 - air + 1e-8) * (gauss * middle)) / torch.maximum(
        torch.sum(middle), torch.tensor([1e-8]))                                      # :46-52
    return (loss_before + loss_middle).reshape(())


def ins_criterion(pred_ins: torch.Tensor, gt_labels: torch.Tensor, ins_num: int):
    """Instance-matching loss (networks/evaluator.py:19-74): one-hot ground truth over the labels present,
    cross-entropy and soft-IoU cost matrices by broadcasting, Hungarian matching (scipy) on the rows of the classes
    present, ``(sum, valid_ce, invalid_ce, valid_siou)``."""
    from scipy.optimize import linear_sum_assignment
    valid = torch.unique(gt_labels)                                                    # :21
    v = len(valid)
    gt = torch.zeros(gt_labels.shape[0], ins_num)
    gt[..., :v] = torch.nn.functional.one_hot(gt_labels.long())[..., valid.long()].float()   # :22-25
    p, g = pred_ins.permute(1, 0)[None, :, :], gt.permute(1, 0)[:, None, :]           # :55-58
    cost_ce = torch.mean(-g * torch.log(p + 1e-8) - (1 - g) * torch.log(1 - p + 1e-8), dim=-1)
    tp = torch.sum(p * g, dim=-1)
    fp = torch.sum(p, dim=-1) - tp
    fn = torch.sum(g, dim=-1) - tp
    cost_siou = 1.0 - tp / (tp + fp + fn + 1e-6)                                       # :63-67
    with torch.no_grad():                                                              # :42-52
        rows, cols = linear_sum_assignment((cost_ce + cost_siou)[:v].numpy())
        rest = np.array(sorted(set(range(ins_num)) - set(cols.tolist())), dtype=np.int64)
    valid_ce = torch.mean(cost_ce[rows, cols])                                         # :27
    invalid_ce = torch.mean(pred_ins[:, rest]) if rest.size else torch.zeros(())       # :29-32
    valid_siou = torch.mean(cost_siou[rows, cols])                                     # :33
    return valid_ce + invalid_ce + valid_siou, valid_ce, invalid_ce, valid_siou


# --------------------------------------------------------------------------------------
# object manipulation (row N3) -- networks/manipulator.py:18-205.  Pinned by G12: the reference's own four function
# definitions, compiled from its source text in a namespace of torch / F / its sample_pdf (the module's top-level
# imports need packages the image lacks; the functions do not), tests/golden/make_golden.py:g12_manipulator
# --------------------------------------------------------------------------------------
def exchanger(ori_raw: torch.Tensor, tar_raws, ori_raw_pred: torch.Tensor, tar_raw_preds, move_labels):
    """Label surgery between an original and a transformed scene.  Returns
    ``(ori_raw_new, tar_raws, ori_pred_label, tar_pred_label_of_last_object)``; unlike the reference the input is
    not edited in place."""
    label = lambda x: torch.argmax(torch.sigmoid(x), dim=-1)
    out = ori_raw.clone()
    ori_point = label(ori_raw[..., 4:])                                            # :19-21
    ori_ray = label(ori_raw_pred[..., :-1])[:, None].expand_as(ori_point)          # :23-26
    tar_point = None
    for tar_raw, tar_pred, move in zip(tar_raws, tar_raw_preds, move_labels):
        ori_point = torch.where((ori_point == move) & (ori_ray != move), ori_ray, ori_point)       # :33-36
        filling = (ori_ray == move) & (ori_point != move)                                          # :40-42
        tar_point = label(tar_raw[..., 4:])                                                        # :45-48
        tar_ray = label(tar_pred[..., :-1])[:, None].expand_as(tar_point)                          # :50-53
        tar_point = torch.where((tar_point == move) & (tar_ray != move), tar_ray, tar_point)       # :57-60
        take = filling | (tar_point == move)           # operation_mask 1 (target only, or both) and the fillings
        wipe = (ori_point == move) & (tar_point != move)                                           # operation_mask 0
        out = torch.where(take[..., None], tar_raw, out)                                           # :78-80
        out = torch.where(wipe[..., None], out * 0, out)                                           # :81
    return out, tar_raws, ori_point, tar_point


def manipulator_depths(n_rays: int, near: float, far: float, n_samples: int) -> torch.Tensor:
    """``z = near*(1-t) + far*t`` (networks/manipulator.py:116-120; not the same rounding as ``z_val_sample``)."""
    lo, hi = near * torch.ones(n_rays, 1), far * torch.ones(n_rays, 1)
    t = torch.linspace(0.0, 1.0, steps=n_samples)
    return (lo * (1.0 - t) + hi * t).expand(n_rays, n_samples)


def manipulator(coarse: Params, fine: Params, ori_rays: torch.Tensor, f_tar_rays, n_samples: int, n_importance: int,
                near: float, far: float, move_labels, draws):
    """Two-pass re-render of a scene with moved objects (networks/manipulator.py:137-205).  ``draws``: the uniform
    draws of the ``2 + len(f_tar_rays)`` stochastic ``sample_pdf`` calls, in call order.  Returns
    ``(final_rgb, final_ins, tar_rgb, tar_ins_accum)`` plus a dict of intermediates for stage-wise checks."""
    draws = list(draws)

    def resample(z, w):
        mid = 0.5 * (z[..., 1:] + z[..., :-1])
        return inverse_cdf(mid, w[..., 1:-1], n_importance, det=False, u=draws.pop(0))

    def accum_labels(rays, z, z_new):
        z_full = merge_depths(z, z_new)
        raw_full = field_on_rays(fine, rays[0], rays[1], z_full)
        return composite(raw_full, z_full, rays[1], manipulator=True)[3], z_full

    n = ori_rays.shape[1]
    ori_z = manipulator_depths(n, near, far, n_samples)
    ori_raw = field_on_rays(coarse, ori_rays[0], ori_rays[1], ori_z)                       # :140-141
    _, ori_w, _, _ = composite(ori_raw, ori_z, ori_rays[1], manipulator=True)              # :144
    ori_acc, z_full_first = accum_labels(ori_rays, ori_z, resample(ori_z, ori_w))          # :147-152
    tar_raws, tar_zs, tar_samples, tar_accs = [], [], [], []
    tar_rgb = tar_acc = None
    for tar_rays in f_tar_rays:                                                            # :155-176
        tar_z = manipulator_depths(n, near, far, n_samples)
        tar_raw = field_on_rays(coarse, tar_rays[0], tar_rays[1], tar_z)
        tar_rgb, tar_w, _, _ = composite(tar_raw, tar_z, tar_rays[1], manipulator=True)
        z_new = resample(tar_z, tar_w)
        tar_acc, _ = accum_labels(tar_rays, tar_z, z_new)
        tar_raws.append(tar_raw); tar_zs.append(tar_z); tar_samples.append(z_new); tar_accs.append(tar_acc)
    ori_raw, _, ex1_ori, ex1_tar = exchanger(ori_raw, tar_raws, ori_acc, tar_accs, move_labels)   # :179
    _, ori_w, _, _ = composite(ori_raw, ori_z, ori_rays[1], manipulator=True)              # :183
    ori_new = resample(ori_z, ori_w)                                                       # :186-187
    extra = torch.cat(tar_samples, -1)                                                     # :190
    z_final = torch.sort(torch.cat([ori_z, ori_new, extra], -1), -1).values                # :191
    ori_raw = field_on_rays(fine, ori_rays[0], ori_rays[1], z_final)                       # :194-195
    for i, tar_rays in enumerate(f_tar_rays):                                              # :192-199
        tar_z_all = torch.sort(torch.cat([tar_zs[i], ori_new, extra], -1), -1).values
        tar_raws[i] = field_on_rays(fine, tar_rays[0], tar_rays[1], tar_z_all)
    ori_raw, _, ex2_ori, ex2_tar = exchanger(ori_raw, tar_raws, ori_acc, tar_accs, move_labels)   # :201
    final_rgb, _, _, final_ins = composite(ori_raw, z_final, ori_rays[1], manipulator=True)       # :203
    trace = {"z_full_first": z_full_first, "z_final": z_final, "ex1_ori_label": ex1_ori, "ex1_tar_label": ex1_tar,
             "ex2_ori_label": ex2_ori, "ex2_tar_label": ex2_tar}
    return final_rgb, final_ins, tar_rgb, tar_acc, trace


# --------------------------------------------------------------------------------------
# training ray selectors
# --------------------------------------------------------------------------------------
def select_full(rgb: torch.Tensor, pose: torch.Tensor, K, labels: torch.Tensor, picked) -> Tuple:
    """``get_select_full`` (networks/helpers.py:99-111) with the host draw passed in: all H*W rays, then the
    picked flat pixel indices.  Returns ``(target_c, target_i, batch_rays[2, N, 3])``."""
    H, W = rgb.shape[:2]
    rays_o, rays_d = rays_from_camera(H, W, K, pose)
    idx = torch.as_tensor(np.asarray(picked), dtype=torch.long)
    rows, cols = idx // W, idx % W
    rays = torch.stack([rays_o[rows, cols], rays_d[rows, cols]], 0)
    return rgb[rows, cols], labels[rows, cols], rays


def select_crop(rgb, pose, K, labels, ins_index, crop_mask, n_train, labeled_pick, unlabeled_pick) -> Tuple:
    """``get_select_crop`` (networks/helpers.py:64-95) with both draws passed in.  Labelled pixels (30 % of the
    batch, at most ``len(ins_index)``) come last; the unlabelled positions index ``crop_indices`` directly, as the
    reference does (:85-86).  Returns ``(target_c, target_i, batch_rays, N_ins)``."""
    H, W = rgb.shape[:2]
    ins_index = np.asarray(ins_index)
    n_ins = min(int(n_train * 0.3), len(ins_index))
    crop_indices = np.where(np.asarray(crop_mask).reshape(-1) == 1)[0]
    labeled = ins_index[np.asarray(labeled_pick)]
    unlabeled = crop_indices[np.asarray(unlabeled_pick)]
    rays_o, rays_d = rays_from_camera(H, W, K, pose)
    sel = torch.as_tensor(np.concatenate([unlabeled, labeled]), dtype=torch.long)
    rows, cols = sel // W, sel % W
    lab = torch.as_tensor(labeled, dtype=torch.long)
    rays = torch.stack([rays_o[rows, cols], rays_d[rows, cols]], 0)
    return rgb[rows, cols], labels[lab // W, lab % W], rays, n_ins
