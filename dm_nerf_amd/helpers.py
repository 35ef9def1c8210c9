"""Ray and sampling helpers with the reference's names and signatures
(networks/helpers.py:50-61,114-155), executed by the gfx950 kernels."""
from __future__ import annotations

from functools import lru_cache

import torch

from . import ops


def _device() -> torch.device:
    if not torch.cuda.is_available():
        raise ops._lib.HipLibraryError("dm_nerf_amd needs a ROCm GPU (no CPU path)")
    return torch.device("cuda", torch.cuda.current_device())


def get_rays_k(H, W, K, c2w):
    """Per-pixel ray origins / directions ``[H, W, 3]`` (networks/helpers.py:50-61).
    ``K`` is the host-side intrinsics array the loaders build; ``c2w`` a 3x4 or 4x4 pose."""
    return ops.raygen(int(H), int(W), K, c2w)


@lru_cache(maxsize=64)
def _lattice(near: float, far: float, n_samples: int, device_index: int) -> torch.Tensor:
    # 64 floats of host arithmetic, evaluated with the same torch CPU ops as the reference
    # (networks/helpers.py:115-118) so the lattice is bit-identical to it, then kept on the device.
    lo = near * torch.ones(size=(1, 1))
    hi = far * torch.ones(size=(1, 1))
    t = torch.linspace(0.0, 1.0, steps=n_samples)
    return (lo + t * (hi - lo)).to(torch.device("cuda", device_index))


def z_val_sample(N_rays, near, far, N_samples):
    """Coarse depth lattice expanded to ``[N_rays, N_samples]`` (networks/helpers.py:114-119);
    like the reference it returns an expanded (stride-0) view."""
    dev = _device()
    return _lattice(float(near), float(far), int(N_samples), dev.index).expand([int(N_rays), int(N_samples)])


@lru_cache(maxsize=16)
def _unit_linspace(n: int, device_index: int) -> torch.Tensor:
    return torch.linspace(0.0, 1.0, steps=n).to(torch.device("cuda", device_index))


def sample_pdf(bins, weights, N_samples, det=False, u=None):
    """Inverse-CDF sampling (networks/helpers.py:123-155).  ``u`` optionally supplies the uniform
    draws (the reference draws them with the global ``torch.rand``)."""
    n = bins.shape[0]
    if det:
        u = _unit_linspace(int(N_samples), bins.device.index)       # one row shared by all rays
    elif u is None:
        u = torch.rand(n, int(N_samples), device=bins.device)
    return ops.sample_pdf(bins, weights, int(N_samples), u)


def get_select_full(rgb, pose, K, ins_target, N_train, selected_index=None):
    """Training batch of ``N_train`` distinct pixels of one image: ``(target_c, target_i, batch_rays)``
    (networks/helpers.py:99-111).  The reference builds all H*W rays on every step and permutes H*W indices
    on the host; here the indices are drawn on the device and only the selected rays are generated.
    ``selected_index`` (flat ``row * W + col``) overrides the draw (the reference uses ``np.random.choice``)."""
    H, W = rgb.shape[:2]
    dev = _device()
    if selected_index is None:
        selected_index = torch.randperm(H * W, device=dev)[: int(N_train)]
    selected_index = torch.as_tensor(selected_index, dtype=torch.int64, device=dev)
    rays_o, rays_d = ops.raygen_select(int(H), int(W), K, pose, selected_index)
    batch_rays = torch.stack([rays_o, rays_d], 0)
    target_c = rgb.reshape(H * W, -1)[selected_index]
    target_i = ins_target.reshape(H * W, *ins_target.shape[2:])[selected_index]
    return target_c, target_i, batch_rays


def get_select_crop(rgb, pose, K, ins_target, ins_index, crop_mask, N_train, labeled_pick=None, unlabeled_pick=None):
    """ScanNet crop sampler: ``(target_c, target_i, batch_rays, N_ins)`` (networks/helpers.py:64-95).
    30 % of the batch (at most ``len(ins_index)``) are labelled pixels and come LAST, so that
    ``ins_fine[-N_ins:]`` in dm_nerf (networks/render.py:88-90) lines up; the rest are drawn from the crop
    (indexing ``crop_indices`` with the drawn positions, as the reference does at helpers.py:85-86).
    ``labeled_pick`` / ``unlabeled_pick`` override the two draws (positions into ``ins_index`` /
    ``crop_indices``)."""
    H, W = rgb.shape[:2]
    dev = _device()
    ins_index = torch.as_tensor(ins_index, dtype=torch.int64, device=dev).reshape(-1)
    n_ins = min(int(N_train * 0.3), ins_index.numel())
    n_rgb = int(N_train) - n_ins
    crop_indices = torch.nonzero(torch.as_tensor(crop_mask, device=dev).reshape(-1) == 1).reshape(-1)
    if labeled_pick is None:
        labeled_pick = torch.randperm(ins_index.numel(), device=dev)[:n_ins]
    labeled_idx = ins_index[torch.as_tensor(labeled_pick, dtype=torch.int64, device=dev)]
    if unlabeled_pick is None:
        # the reference draws positions among the unlabelled crop pixels (helpers.py:84-85)
        n_unlabeled = crop_indices.numel() - int(torch.isin(crop_indices, labeled_idx).sum())
        unlabeled_pick = torch.randperm(n_unlabeled, device=dev)[:n_rgb]
    unlabeled_idx = crop_indices[torch.as_tensor(unlabeled_pick, dtype=torch.int64, device=dev)]
    selected = torch.cat([unlabeled_idx, labeled_idx], 0)
    rays_o, rays_d = ops.raygen_select(int(H), int(W), K, pose, selected)
    batch_rays = torch.stack([rays_o, rays_d], 0)
    target_c = rgb.reshape(H * W, -1)[selected]
    target_i = ins_target.reshape(H * W, *ins_target.shape[2:])[labeled_idx]
    return target_c, target_i, batch_rays, n_ins
