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
