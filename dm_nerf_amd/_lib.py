"""ctypes binding of the C ABI declared in ``include/dmnerf_hip.h``.

There is deliberately no fallback: if the HIP library is missing or a call fails, the error
propagates (``HipLibraryError``).  The product never routes through a CPU implementation."""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_long, c_void_p
from typing import Optional

# DMNERF_LIB points at an alternative build of the same ABI (timing-only ablation builds, tools/ablate.sh)
LIB_PATH = os.environ.get("DMNERF_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc",
                                                        "libdmnerf_hip.so")

OK = 0
ABI_VERSION = 1


class HipLibraryError(RuntimeError):
    pass


_P = c_void_p
# name -> (restype, argtypes); mirrors include/dmnerf_hip.h one to one
SIGNATURES = {
    "dmn_abi_version": (c_int, []),
    "dmn_error_string": (c_char_p, [c_int]),
    "dmn_device_name": (c_int, [c_char_p, c_int]),
    "dmn_field_param_floats": (c_int, [c_int]),
    "dmn_field_blob_floats": (c_long, [c_int]),
    "dmn_field_pack_index": (c_int, [c_int, _P]),
    "dmn_field_pack": (c_int, [_P, _P, c_long, _P, _P]),
    "dmn_raygen": (c_int, [c_int, c_int, c_float, c_float, c_float, c_float, c_float, _P, _P, _P, _P]),
    "dmn_raygen_select": (c_int, [c_int, c_int, c_float, c_float, c_float, c_float, c_float, _P, _P, c_long, _P, _P, _P]),
    "dmn_jitter_depths": (c_int, [_P, c_long, _P, c_long, c_int, _P, _P]),
    "dmn_encode": (c_int, [_P, c_long, c_int, _P, _P]),
    "dmn_field_fwd_rays": (c_int, [_P, _P, _P, c_long, c_long, c_int, _P, c_int, _P, _P]),
    "dmn_field_fwd_embedded": (c_int, [_P, c_long, _P, c_int, _P, _P]),
    "dmn_field_blob_x3_bytes": (c_long, [c_int]),
    "dmn_field_pack_index_x3_len": (c_long, [c_int]),
    "dmn_field_pack_index_x3": (c_int, [c_int, _P]),
    "dmn_field_pack_x3": (c_int, [_P, _P, c_int, _P, _P]),
    "dmn_field_fwd_rays_x3": (c_int, [_P, _P, _P, c_long, c_long, c_int, _P, c_int, c_int, _P, _P, _P]),
    "dmn_field_fwd_rays_x3_sigma": (c_int, [_P, _P, _P, c_long, c_long, c_int, _P, c_int, c_int, _P, _P, _P]),
    "dmn_field_fwd_embedded_x3": (c_int, [_P, c_long, _P, c_int, c_int, _P, _P, _P]),
    "dmn_field_blob_x6_bytes": (c_long, [c_int]),
    "dmn_field_pack_index_x6_len": (c_long, [c_int]),
    "dmn_field_pack_index_x6": (c_int, [c_int, _P]),
    "dmn_field_pack_x6": (c_int, [_P, _P, c_int, _P, _P]),
    "dmn_field_fwd_rays_x6": (c_int, [_P, _P, _P, c_long, c_long, c_int, _P, c_int, c_int, _P, _P]),
    "dmn_field_fwd_embedded_x6": (c_int, [_P, c_long, _P, c_int, _P, _P]),
    "dmn_field_fwd_rays_x6_save": (c_int, [_P, _P, _P, c_long, c_long, c_int, _P, c_int, _P, _P, _P]),
    "dmn_field_composite_x3": (c_int, [_P, _P, _P, c_long, c_long, c_int, _P, c_int, c_int, c_int, _P, _P, _P, _P, _P, c_long,
                                       c_int, _P, _P, _P]),
    "dmn_render_fused": (c_int, [_P, _P, _P, c_long, _P, c_long, c_long, c_int, c_int, _P, _P, c_int, c_int, c_int, _P, _P,
                                 _P, _P, _P, _P, _P, _P, _P]),
    "dmn_render_workspace_floats": (c_long, [c_long, c_int, c_int, c_int]),
    "dmn_render_rays": (c_int, [_P, _P, _P, c_long, _P, _P, c_long, c_long, c_int, c_int, _P, _P, c_int, c_int, _P, _P, _P,
                                _P, _P, _P, _P, _P, _P]),
    "dmn_sigmoid_argmax": (c_int, [_P, c_long, c_long, c_int, _P, _P]),
    "dmn_exchange": (c_int, [_P, _P, _P, _P, _P, _P, c_long, c_int, c_int, c_int, _P]),
    "dmn_field_save_floats": (c_long, [c_long]),
    "dmn_field_grad_floats": (c_long, [c_long, c_int]),
    "dmn_field_fwd_rays_save": (c_int, [_P, _P, _P, c_long, c_long, c_int, _P, c_int, _P, _P, _P]),
    "dmn_field_fwd_rays_x3_save": (c_int, [_P, _P, _P, c_long, c_long, c_int, _P, c_int, c_int, _P, _P, _P]),
    "dmn_field_bwd_blob_floats": (c_long, [c_int]),
    "dmn_field_bwd_pack_index": (c_int, [c_int, _P]),
    "dmn_field_bwd_blob_x3_bytes": (c_long, [c_int]),
    "dmn_field_bwd_pack_index_x3_len": (c_long, [c_int]),
    "dmn_field_bwd_pack_index_x3": (c_int, [c_int, _P]),
    "dmn_field_bwd_data_x3": (c_int, [_P, _P, _P, c_long, c_int, _P, _P]),
    "dmn_field_pack_index_x6w": (c_int, [c_int, _P]),
    "dmn_field_fwd_rays_x6w": (c_int, [_P, _P, _P, c_long, c_long, c_int, _P, c_int, c_int, _P, _P]),
    "dmn_field_fwd_embedded_x6w": (c_int, [_P, c_long, _P, c_int, _P, _P]),
    "dmn_field_bwd_blob_x6_bytes": (c_long, [c_int]),
    "dmn_field_bwd_pack_index_x6_len": (c_long, [c_int]),
    "dmn_field_bwd_pack_index_x6": (c_int, [c_int, _P]),
    "dmn_field_pack_x6_n": (c_int, [_P, _P, c_long, _P, _P]),
    "dmn_field_bwd_data_x6": (c_int, [_P, _P, _P, c_long, c_int, _P, _P]),
    "dmn_field_pack_x3_n": (c_int, [_P, _P, c_long, c_int, _P, _P]),
    "dmn_field_wgrad_index_len": (c_long, [c_int]),
    "dmn_field_wgrad_index": (c_int, [c_int, _P]),
    "dmn_field_bwd_data": (c_int, [_P, _P, _P, c_long, c_int, _P, _P]),
    "dmn_field_wgrad_workspace_floats": (c_long, [c_int, c_int]),
    "dmn_field_bwd_weights": (c_int, [_P, _P, c_long, c_int, _P, c_int, c_int, _P, _P, c_int, _P]),
    "dmn_penalizer_fwd": (c_int, [_P, _P, c_long, _P, _P, c_long, c_int, c_int, c_float, c_float, _P, _P]),
    "dmn_penalizer_loss": (c_int, [_P, c_int, _P, _P]),
    "dmn_penalizer_bwd": (c_int, [_P, _P, c_long, _P, _P, c_long, c_int, c_int, c_float, c_float, _P, _P, _P, _P]),
    "dmn_ins_stats_fwd": (c_int, [_P, _P, _P, c_long, c_int, c_int, _P, _P]),
    "dmn_ins_stats_bwd": (c_int, [_P, _P, c_long, c_int, c_int, _P, _P, _P]),
    "dmn_ins_label_prep": (c_int, [_P, c_long, c_int, _P, _P, _P, _P]),
    "dmn_ins_label_hist": (c_int, [_P, c_long, c_int, _P, _P, _P]),
    "dmn_ins_label_slots": (c_int, [_P, c_long, _P, c_int, c_int, _P, _P, _P, _P]),
    "dmn_ins_stats_sum": (c_int, [_P, c_long, c_int, c_int, _P, _P]),
    "dmn_ins_stats_chunks": (c_int, [c_long]),
    "dmn_ins_stats_scan": (c_int, [_P, _P, c_long, c_int, c_int, _P, _P]),
    "dmn_ins_cost": (c_int, [_P, _P, _P, _P, c_long, _P, c_int, c_int, _P, _P, _P, _P]),
    "dmn_ins_assign": (c_int, [_P, _P, c_int, c_int, _P, c_int, _P]),
    "dmn_ins_loss": (c_int, [_P, _P, _P, _P, _P, c_long, _P, c_int, c_int, _P, _P]),
    "dmn_ins_loss_bwd": (c_int, [_P, _P, _P, _P, _P, c_long, _P, c_int, c_int, _P, _P, _P, _P, _P, _P]),
    "dmn_composite_fwd": (c_int, [_P, _P, c_long, _P, c_long, c_int, c_int, c_int, _P, _P, _P, _P, _P]),
    "dmn_composite_bwd": (c_int, [_P, _P, c_long, _P, _P, _P, _P, _P, _P, c_long, c_int, c_int, c_int, _P, _P]),
    "dmn_sample_pdf": (c_int, [_P, c_long, _P, c_long, _P, c_long, c_long, c_int, c_int, _P, _P]),
    "dmn_resample_merge": (c_int, [_P, c_long, _P, _P, c_long, c_long, c_int, c_int, _P, _P]),
    "dmn_merge_depths": (c_int, [_P, c_long, c_int, _P, c_long, c_int, c_long, _P, _P]),
}

_lib: Optional[ctypes.CDLL] = None


def load() -> ctypes.CDLL:
    """Load (once) and type the shared library.  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryError(
            f"{LIB_PATH} is missing: build it with `python -m dm_nerf_amd.build` "
            "(dm_nerf_amd has no CPU fallback)")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype, fn.argtypes = res, args
    if lib.dmn_abi_version() != ABI_VERSION:
        raise HipLibraryError(f"ABI version {lib.dmn_abi_version()} != expected {ABI_VERSION}; rebuild")
    _lib = lib
    return lib


def check(code: int, what: str) -> None:
    if code != OK:
        msg = load().dmn_error_string(code).decode()
        raise HipLibraryError(f"{what}: {msg} (code {code})")


def ptr(t) -> Optional[int]:
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def current_stream() -> int:
    import torch
    return torch.cuda.current_stream().cuda_stream
