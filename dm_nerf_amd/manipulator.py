"""The two kernels the reference's object manipulator runs outside ``dm_nerf``
(networks/manipulator.py:86-134).  The label surgery (``exchanger``) and the demo / evaluation
drivers around them are out of scope."""
from __future__ import annotations

import torch

from . import ops
from .helpers import _lattice
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
