"""dm_nerf_amd -- MI355X-native (gfx950) volumetric-render hot path behind DM-NeRF's Python API.

Exports the reference's names (SURVEY.md name map): ``create_nerf``, ``get_embedder``,
``DM_NeRF``, ``dm_nerf``, ``render_train``, ``render_test``, ``sample_pdf``, ``z_val_sample``,
``get_rays_k``, ``manipulator_nerf``, ``manipulator_render``; plus the nerf-pytorch aliases the
north-star uses (``render``, ``raw2outputs``, ``render_path``, ``get_rays``-style names).

All compute runs in hand-written HIP kernels loaded from ``csrc/libdmnerf_hip.so`` through the
C ABI of ``include/dmnerf_hip.h``.  There is no CPU fallback: without the library or without a
ROCm device the calls raise."""
from ._lib import HipLibraryError, LIB_PATH, load as load_library
from .config import create_nerf, default_args
from .dm_nerf import DM_NeRF, Embedder, get_embedder
from .evaluator import img2mse, ins_criterion, ins_criterion_many, mse2psnr
from .helpers import get_rays_k, get_select_crop, get_select_full, sample_pdf, z_val_sample
from .manipulator import exchanger, manipulator, manipulator_nerf, manipulator_render
from .ops import DEFAULT_PRECISION, PRECISIONS
from .penalizer import emptiness_penalizer, ins_penalizer
from .render import dm_nerf, render_train
from .tester import GraphedImage, render_image, render_test
from . import autograd, parallel, synthetic, trainer

# nerf-pytorch names used by BASELINE.json's north_star
render = dm_nerf
raw2outputs = render_train
render_path = render_test

__all__ = [
    "HipLibraryError", "LIB_PATH", "load_library", "create_nerf", "default_args", "DM_NeRF", "Embedder",
    "get_embedder", "emptiness_penalizer", "ins_penalizer", "ins_criterion", "ins_criterion_many", "img2mse", "mse2psnr", "get_rays_k", "get_select_full", "get_select_crop", "sample_pdf", "z_val_sample", "manipulator_nerf", "manipulator_render", "exchanger", "manipulator",
    "dm_nerf", "render_train", "render_image", "GraphedImage", "render_test", "render", "raw2outputs", "render_path", "parallel",
    "trainer", "DEFAULT_PRECISION", "PRECISIONS",
]
