"""Model factory with the reference's name and return tuple (config.py:146-159).

The reference's flag parser / run-directory setup (config.py:9-143,162-188) is control plane and
out of scope; ``default_args`` provides a namespace carrying the hot-path defaults those flags
define, for callers that do not bring their own."""
from __future__ import annotations

from types import SimpleNamespace

import torch

from .dm_nerf import DM_NeRF, get_embedder


def default_args(**overrides) -> SimpleNamespace:
    """Hot-path defaults (config.py:23-59 and configs/test/dmsr/study.txt:7-14)."""
    args = SimpleNamespace(netdepth=8, netwidth=256, multires=10, multires_views=4, i_embed=0,
                           N_samples=64, N_importance=128, N_train=3072, N_test=4096, perturb=0.0,
                           near=4.0, far=15.0, ins_num=13, is_train=False, N_ins=None,
                           device=torch.device("cuda", torch.cuda.current_device())
                           if torch.cuda.is_available() else torch.device("cpu"))
    for k, v in overrides.items():
        setattr(args, k, v)
    return args


def create_nerf(args):
    """``(position_embedder, view_embedder, model_coarse, model_fine, args)`` (config.py:146-159)."""
    position_embedder, input_ch_pos = get_embedder(args.multires, args.i_embed)
    view_embedder, input_ch_view = get_embedder(args.multires_views, args.i_embed)
    model_coarse = DM_NeRF(args.netdepth, args.netwidth, input_ch_pos, input_ch_view, [4], args.ins_num).to(args.device)
    model_fine = DM_NeRF(args.netdepth, args.netwidth, input_ch_pos, input_ch_view, [4], args.ins_num).to(args.device)
    return position_embedder, view_embedder, model_coarse, model_fine, args
