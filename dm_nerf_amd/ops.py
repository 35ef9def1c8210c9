"""Tensor-level wrappers over the C ABI: argument checks on the host (shapes, dtype, device,
strides), output allocation with torch, launch on torch's current HIP stream.

Every function requires ROCm-device tensors; nothing here computes on the CPU."""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import check, current_stream, load, ptr

MAX_INS_CHANNELS = 128

# arithmetic of the field kernel (include/dmnerf_hip.h DMN_PREC_*)
# "bf16x6_32": the bf16x6 arithmetic on the 32x32x16 kernel that also has a saving (training) and a fused variant; plain
# "bf16x6" runs inference on the 16x16x32 kernel where that exists (uses_x6w) -- same arithmetic, other summation order
PRECISIONS = {"fp32": 0, "fp16x3": 1, "bf16x3": 2, "bf16x6": 3, "bf16x6_32": 3}
X6_PRECISIONS = ("bf16x6", "bf16x6_32")
X6W_MAX_CHANNELS = 128    # ins_num + 1 up to which the 16x16x32 inference kernel is used (csrc/field_fwd_x6w.hip): all of them


def uses_x6w(ins_num: int, precision: str = "bf16x6") -> bool:
    """True when inference in `precision` runs on the v_mfma_f32_16x16x32_bf16 kernel (and reads the "x6w" blob)."""
    return precision == "bf16x6" and ins_num + 1 <= X6W_MAX_CHANNELS


def inference_precision_code(precision: str, ins_num: int) -> int:
    """`precision` argument of the C entry points that run the inference chain (dmn_render_rays) for the blob
    ``DM_NeRF.blob(precision)``: DMN_PREC_BF16X6W where bf16x6 inference runs on the 16x16x32 kernel."""
    return 4 if uses_x6w(ins_num, precision) else precision_code(precision)


def save_blob_name(precision: str) -> str:
    """DM_NeRF.blob() name of the blob the SAVING forward and the fused kernels of `precision` read."""
    return "bf16x6_32" if precision in X6_PRECISIONS else precision
X3_PRECISIONS = ("fp16x3", "bf16x3")        # two-way split kernels (22 / 16 operand bits); "bf16x6" carries all 24
# What dm_nerf(), the image loop and DM_NeRF.forward() run when the caller does not say (args.precision / model.precision):
# fp32-grade products on the bf16 matrix pipe -- every fp32 operand enters the products with all of its 24 significand
# bits (exact three-way bf16 splits, six MFMA products, fp32 accumulation).  Measured against an fp64 evaluation of the
# field (bench.py `arithmetic_check`): 6.0e-7, where the exact fp32-MFMA kernel is at 6.5e-7 and the reference's own fp32
# result at 4.3e-7 -- fp32-grade, not bit-exact fp32; 1.87x the exact kernel's rays/s.  "fp32" is the parity mode (exact
# fp32 MFMA products, a k-ordered fma chain); "fp16x3" / "bf16x3" are the narrower two-way splits (opt-in).
DEFAULT_PRECISION = "bf16x6"
STATUS_FP16_RANGE = 1
_status = {}


def precision_code(precision: str) -> int:
    try:
        return PRECISIONS[precision]
    except KeyError:
        raise ValueError(f"precision must be one of {sorted(PRECISIONS)}, got {precision!r}") from None


def status_word(device) -> torch.Tensor:
    """Per-device sticky status int the split-precision kernels OR flags into."""
    key = torch.device(device).index
    if key not in _status:
        _status[key] = torch.zeros(1, dtype=torch.int32, device=device)
    return _status[key]


def take_status(device) -> int:
    """Read and clear the status word (synchronises with the device)."""
    w = status_word(device)
    v = int(w.item())
    if v:
        w.zero_()
    return v

# Optional per-launch timing hook (bench.py): when set to a list, every field / composite launch
# appends (kernel_name, start_event, end_event, work_items) with HIP events recorded on the launch
# stream.  None (the default) adds nothing to the launch path.
PROFILE = None


class _timed:
    def __init__(self, name: str, work: int):
        self.name, self.work = name, work

    def __enter__(self):
        if PROFILE is not None:
            self.t0 = torch.cuda.Event(enable_timing=True)
            self.t1 = torch.cuda.Event(enable_timing=True)
            self.t0.record()
        return self

    def __exit__(self, *exc):
        if PROFILE is not None:
            self.t1.record()
            PROFILE.append((self.name, self.t0, self.t1, self.work))
        return False


def _dev(t: torch.Tensor, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise _lib.HipLibraryError(f"{name} is on {t.device}: dm_nerf_amd runs on a ROCm GPU only (no CPU path)")
    if t.dtype != torch.float32:
        raise TypeError(f"{name} must be float32, got {t.dtype}")
    return t


def _rows(z: torch.Tensor, n_rays: int, name: str) -> Tuple[torch.Tensor, int]:
    """Depth-like [N,S] tensor -> (tensor to point at, ray stride).  An expanded lattice
    (stride 0, as z_val_sample returns) is passed as one shared row."""
    _dev(z, name)
    if z.dim() != 2 or z.shape[0] != n_rays:
        raise ValueError(f"{name} must be [{n_rays}, S], got {tuple(z.shape)}")
    if n_rays > 1 and z.stride(0) == 0 and z.stride(1) == 1:
        return z, 0
    z = z.contiguous()
    return z, z.shape[1]


def _draws(u: Optional[torch.Tensor], n_rays: int, n_new: int):
    """Uniform draws: None (kernel evaluates linspace), a shared [n_new] row, or [N, n_new]."""
    if u is None or n_new == 0:
        return None, 0
    _dev(u, "u")
    if u.dim() == 1:
        if u.shape[0] != n_new:
            raise ValueError(f"u must have {n_new} entries")
        return u.contiguous(), 0
    if tuple(u.shape) != (n_rays, n_new):
        raise ValueError(f"u must be [{n_rays}, {n_new}]")
    if n_rays > 1 and u.stride(0) == 0 and u.stride(1) == 1:
        return u, 0
    return u.contiguous(), n_new


# ----------------------------------------------------------------------------------------
def pack_index(ins_num: int, device) -> torch.Tensor:
    lib = load()
    n = lib.dmn_field_blob_floats(ins_num)
    if n < 0:
        raise ValueError(f"ins_num={ins_num} unsupported (ins_num + 1 must be <= {MAX_INS_CHANNELS})")
    idx = np.empty(n, dtype=np.int32)
    check(lib.dmn_field_pack_index(ins_num, idx.ctypes.data), "dmn_field_pack_index")
    return torch.from_numpy(idx).to(device)


def pack_blob(flat_params: torch.Tensor, index: torch.Tensor) -> torch.Tensor:
    _dev(flat_params, "flat_params")
    blob = torch.empty(index.numel(), dtype=torch.float32, device=flat_params.device)
    check(load().dmn_field_pack(ptr(flat_params), ptr(index), index.numel(), ptr(blob), current_stream()),
          "dmn_field_pack")
    return blob


def pack_index_x3(ins_num: int, device) -> torch.Tensor:
    lib = load()
    n = lib.dmn_field_pack_index_x3_len(ins_num)
    if n < 0:
        raise ValueError(f"ins_num={ins_num} unsupported (ins_num + 1 must be <= {MAX_INS_CHANNELS})")
    idx = np.empty(n, dtype=np.int32)
    check(lib.dmn_field_pack_index_x3(ins_num, idx.ctypes.data), "dmn_field_pack_index_x3")
    return torch.from_numpy(idx).to(device)


def pack_blob_x3(flat_params: torch.Tensor, index: torch.Tensor, ins_num: int, precision: str) -> torch.Tensor:
    _dev(flat_params, "flat_params")
    n_bytes = load().dmn_field_blob_x3_bytes(ins_num)
    blob = torch.empty(n_bytes, dtype=torch.uint8, device=flat_params.device)
    check(load().dmn_field_pack_x3(ptr(flat_params), ptr(index), precision_code(precision), ptr(blob),
                                   current_stream()), "dmn_field_pack_x3")
    return blob


def pack_index_x6(ins_num: int, device) -> torch.Tensor:
    lib = load()
    n = lib.dmn_field_pack_index_x6_len(ins_num)
    if n < 0:
        raise ValueError(f"ins_num={ins_num} unsupported (ins_num + 1 must be <= {MAX_INS_CHANNELS})")
    idx = np.empty(n, dtype=np.int32)
    check(lib.dmn_field_pack_index_x6(ins_num, idx.ctypes.data), "dmn_field_pack_index_x6")
    return torch.from_numpy(idx).to(device)


def pack_index_x6w(ins_num: int, device) -> torch.Tensor:
    lib = load()
    n = lib.dmn_field_pack_index_x6_len(ins_num)          # same length: the x6w blob holds the same bytes in another order
    if n < 0:
        raise ValueError(f"ins_num={ins_num} unsupported (ins_num + 1 must be <= {MAX_INS_CHANNELS})")
    idx = np.empty(n, dtype=np.int32)
    check(lib.dmn_field_pack_index_x6w(ins_num, idx.ctypes.data), "dmn_field_pack_index_x6w")
    return torch.from_numpy(idx).to(device)


def pack_blob_x6(flat_params: torch.Tensor, index: torch.Tensor, ins_num: int) -> torch.Tensor:
    """Blob of the full-precision bf16 kernel: every weight as its exact hi / mid / lo bf16 planes."""
    _dev(flat_params, "flat_params")
    n_bytes = load().dmn_field_blob_x6_bytes(ins_num)
    blob = torch.empty(n_bytes, dtype=torch.uint8, device=flat_params.device)
    check(load().dmn_field_pack_x6(ptr(flat_params), ptr(index), ins_num, ptr(blob), current_stream()),
          "dmn_field_pack_x6")
    return blob


def raygen(H: int, W: int, K, c2w) -> Tuple[torch.Tensor, torch.Tensor]:
    """get_rays_k; K: 3x3/4x4 array-like (host), c2w: 3x4/4x4 (host array or tensor)."""
    K = np.asarray(K, dtype=np.float64)
    c2w_np = (c2w.detach().cpu().numpy() if torch.is_tensor(c2w) else np.asarray(c2w)).astype(np.float32)
    c2w_np = np.ascontiguousarray(c2w_np[:3, :4])
    dev = torch.device("cuda", torch.cuda.current_device())
    rays_o = torch.empty(H, W, 3, dtype=torch.float32, device=dev)
    rays_d = torch.empty(H, W, 3, dtype=torch.float32, device=dev)
    f32 = lambda v: float(np.float32(v))
    check(load().dmn_raygen(H, W, f32(K[0, 0]), f32(K[0, 2]), f32(K[1, 1]), f32(K[1, 2]), f32(K[2, 2]),
                            c2w_np.ctypes.data, ptr(rays_o), ptr(rays_d), current_stream()), "dmn_raygen")
    return rays_o, rays_d


def raygen_select(H: int, W: int, K, c2w, pixel_index: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Rays of the selected flat pixel indices only (int64 tensor on the device)."""
    K = np.asarray(K, dtype=np.float64)
    c2w_np = (c2w.detach().cpu().numpy() if torch.is_tensor(c2w) else np.asarray(c2w)).astype(np.float32)
    c2w_np = np.ascontiguousarray(c2w_np[:3, :4])
    if not pixel_index.is_cuda or pixel_index.dtype != torch.int64:
        raise TypeError("pixel_index must be an int64 tensor on the ROCm device")
    pixel_index = pixel_index.contiguous()
    n = pixel_index.numel()
    rays_o = torch.empty(n, 3, dtype=torch.float32, device=pixel_index.device)
    rays_d = torch.empty(n, 3, dtype=torch.float32, device=pixel_index.device)
    f32 = lambda v: float(np.float32(v))
    check(load().dmn_raygen_select(H, W, f32(K[0, 0]), f32(K[0, 2]), f32(K[1, 1]), f32(K[1, 2]), f32(K[2, 2]),
                                   c2w_np.ctypes.data, ptr(pixel_index), n, ptr(rays_o), ptr(rays_d),
                                   current_stream()), "dmn_raygen_select")
    return rays_o, rays_d


def jitter_depths(z: torch.Tensor, t_rand: torch.Tensor) -> torch.Tensor:
    n, s = z.shape
    zz, stride = _rows(z, n, "z_vals")
    t_rand = _dev(t_rand, "t_rand").contiguous()
    out = torch.empty(n, s, dtype=torch.float32, device=z.device)
    check(load().dmn_jitter_depths(ptr(zz), stride, ptr(t_rand), n, s, ptr(out), current_stream()),
          "dmn_jitter_depths")
    return out


def encode(x: torch.Tensor, n_freqs: int) -> torch.Tensor:
    _dev(x, "inputs")
    if x.shape[-1] != 3:
        raise ValueError("encoder input must have 3 coordinates")
    flat = x.reshape(-1, 3).contiguous()
    out = torch.empty(flat.shape[0], 3 + 6 * n_freqs, dtype=torch.float32, device=x.device)
    check(load().dmn_encode(ptr(flat), flat.shape[0], n_freqs, ptr(out), current_stream()), "dmn_encode")
    return out.reshape(*x.shape[:-1], out.shape[-1])


def field_rays(rays_o: torch.Tensor, rays_d: torch.Tensor, z: torch.Tensor, blob: torch.Tensor,
               ins_num: int, precision: str = "fp32", sigma_only: bool = False) -> torch.Tensor:
    """raw [N,S,5+ins_num] of the field on ``o + d*z``.  ``sigma_only`` (bf16 / fp16 matrix-pipe kernels): evaluate the
    trunk and the density head only -- raw[..., 3] is filled, every other channel is zero."""
    if sigma_only and precision == "fp32":
        raise ValueError("sigma_only is a mode of the 16-bit matrix-pipe kernels ('fp16x3' / 'bf16x3' / 'bf16x6')")
    n = rays_o.shape[0]
    rays_o = _dev(rays_o, "rays_o").contiguous()
    rays_d = _dev(rays_d, "rays_d").contiguous()
    zz, stride = _rows(z, n, "z_vals")
    s = z.shape[1]
    raw = torch.empty(n, s, 5 + ins_num, dtype=torch.float32, device=rays_o.device)
    with _timed("field_fwd", n * s):
        if precision == "fp32":
            check(load().dmn_field_fwd_rays(ptr(rays_o), ptr(rays_d), ptr(zz), stride, n, s, ptr(blob), ins_num,
                                            ptr(raw), current_stream()), "dmn_field_fwd_rays")
        elif precision in X6_PRECISIONS:
            fn = load().dmn_field_fwd_rays_x6w if uses_x6w(ins_num, precision) else load().dmn_field_fwd_rays_x6
            check(fn(ptr(rays_o), ptr(rays_d), ptr(zz), stride, n, s, ptr(blob), ins_num, int(sigma_only), ptr(raw),
                     current_stream()), "dmn_field_fwd_rays_x6")
        else:
            fn = load().dmn_field_fwd_rays_x3_sigma if sigma_only else load().dmn_field_fwd_rays_x3
            check(fn(ptr(rays_o), ptr(rays_d), ptr(zz), stride, n, s, ptr(blob), ins_num, precision_code(precision),
                     ptr(raw), ptr(status_word(raw.device)), current_stream()), "dmn_field_fwd_rays_x3")
    return raw


# ---- training: forward with save record, backward-data + weight-gradient kernels ----------------
def bwd_pack_index(ins_num: int, device) -> torch.Tensor:
    lib = load()
    n = lib.dmn_field_bwd_blob_floats(ins_num)
    if n < 0:
        raise ValueError(f"ins_num={ins_num} unsupported (ins_num + 1 must be <= {MAX_INS_CHANNELS})")
    idx = np.empty(n, dtype=np.int32)
    check(lib.dmn_field_bwd_pack_index(ins_num, idx.ctypes.data), "dmn_field_bwd_pack_index")
    return torch.from_numpy(idx).to(device)


def bwd_pack_index_x3(ins_num: int, device) -> torch.Tensor:
    lib = load()
    n = lib.dmn_field_bwd_pack_index_x3_len(ins_num)
    if n < 0:
        raise ValueError(f"ins_num={ins_num} unsupported (ins_num + 1 must be <= {MAX_INS_CHANNELS})")
    idx = np.empty(n, dtype=np.int32)
    check(lib.dmn_field_bwd_pack_index_x3(ins_num, idx.ctypes.data), "dmn_field_bwd_pack_index_x3")
    return torch.from_numpy(idx).to(device)


def pack_bwd_blob_x3(flat_params: torch.Tensor, index: torch.Tensor, ins_num: int) -> torch.Tensor:
    _dev(flat_params, "flat_params")
    lib = load()
    blob = torch.empty(lib.dmn_field_bwd_blob_x3_bytes(ins_num), dtype=torch.uint8, device=flat_params.device)
    check(lib.dmn_field_pack_x3_n(ptr(flat_params), ptr(index), index.numel(), PRECISIONS["bf16x3"], ptr(blob),
                                  current_stream()), "dmn_field_pack_x3_n")
    return blob


def bwd_pack_index_x6(ins_num: int, device) -> torch.Tensor:
    lib = load()
    n = lib.dmn_field_bwd_pack_index_x6_len(ins_num)
    if n < 0:
        raise ValueError(f"ins_num={ins_num} unsupported (ins_num + 1 must be <= {MAX_INS_CHANNELS})")
    idx = np.empty(n, dtype=np.int32)
    check(lib.dmn_field_bwd_pack_index_x6(ins_num, idx.ctypes.data), "dmn_field_bwd_pack_index_x6")
    return torch.from_numpy(idx).to(device)


def pack_bwd_blob_x6(flat_params: torch.Tensor, index: torch.Tensor, ins_num: int) -> torch.Tensor:
    _dev(flat_params, "flat_params")
    lib = load()
    blob = torch.empty(lib.dmn_field_bwd_blob_x6_bytes(ins_num), dtype=torch.uint8, device=flat_params.device)
    check(lib.dmn_field_pack_x6_n(ptr(flat_params), ptr(index), index.numel(), ptr(blob), current_stream()),
          "dmn_field_pack_x6_n")
    return blob


def wgrad_index(ins_num: int, device) -> torch.Tensor:
    lib = load()
    n = lib.dmn_field_wgrad_index_len(ins_num)
    if n < 0:
        raise ValueError(f"ins_num={ins_num} unsupported (ins_num + 1 must be <= {MAX_INS_CHANNELS})")
    idx = np.empty(n, dtype=np.int32)
    check(lib.dmn_field_wgrad_index(ins_num, idx.ctypes.data), "dmn_field_wgrad_index")
    return torch.from_numpy(idx).to(device)


def field_rays_save(rays_o: torch.Tensor, rays_d: torch.Tensor, z: torch.Tensor, blob: torch.Tensor,
                    ins_num: int, precision: str = "fp32") -> Tuple[torch.Tensor, torch.Tensor]:
    """Field forward that also writes the save record the backward kernels read: exact fp32 ("fp32"), fp32-grade products
    on the bf16 pipe ("bf16x6") or three bf16 MFMAs per product ("bf16x3"), each with the blob of that precision."""
    if precision not in ("fp32", "bf16x6", "bf16x6_32", "bf16x3"):
        raise ValueError("the saving forward runs in 'fp32', 'bf16x6' or 'bf16x3'")
    n = rays_o.shape[0]
    rays_o = _dev(rays_o, "rays_o").contiguous()
    rays_d = _dev(rays_d, "rays_d").contiguous()
    zz, stride = _rows(z, n, "z_vals")
    s = z.shape[1]
    raw = torch.empty(n, s, 5 + ins_num, dtype=torch.float32, device=rays_o.device)
    save = torch.empty(load().dmn_field_save_floats(n * s), dtype=torch.float32, device=rays_o.device)
    with _timed("field_fwd_save", n * s):
        if precision == "fp32":
            check(load().dmn_field_fwd_rays_save(ptr(rays_o), ptr(rays_d), ptr(zz), stride, n, s, ptr(blob), ins_num,
                                                 ptr(raw), ptr(save), current_stream()), "dmn_field_fwd_rays_save")
        elif precision in X6_PRECISIONS:        # blob: DM_NeRF.blob(save_blob_name(precision))
            check(load().dmn_field_fwd_rays_x6_save(ptr(rays_o), ptr(rays_d), ptr(zz), stride, n, s, ptr(blob), ins_num,
                                                    ptr(raw), ptr(save), current_stream()), "dmn_field_fwd_rays_x6_save")
        else:
            check(load().dmn_field_fwd_rays_x3_save(ptr(rays_o), ptr(rays_d), ptr(zz), stride, n, s, ptr(blob),
                                                    ins_num, precision_code(precision), ptr(raw), ptr(save),
                                                    current_stream()), "dmn_field_fwd_rays_x3_save")
    return raw, save


BACKWARD_PRECISIONS = ("fp32", "bf16x6", "bf16x3")
BWD_BLOB = {"fp32": "bwd", "bf16x6": "bwd_x6", "bf16x3": "bwd_x3"}        # DM_NeRF.blob() name of the transposed-weight blob


def backward_precision(forward_precision: str) -> str:
    """Arithmetic of the backward kernels (backward-data chain and weight-gradient GEMMs) that goes with a forward: the
    reference back-propagates in fp32 (train_dmsr.py:62-64), so the fp32-grade forwards get fp32-grade gradients --
    "fp32" -> the exact-fp32 MFMA kernels, "bf16x6" -> six bf16 products on exact three-way splits -- and only the
    two-way split forwards ("bf16x3" / "fp16x3", an explicit opt-in) train with 16-operand-bit gradients."""
    if forward_precision in ("fp32", "bf16x6"):
        return forward_precision
    if forward_precision in ("bf16x3", "fp16x3"):
        return "bf16x3"
    raise ValueError(f"precision must be one of {sorted(PRECISIONS)}, got {forward_precision!r}")


def field_backward(g_raw: torch.Tensor, save: torch.Tensor, bwd_blob: torch.Tensor, ins_num: int,
                   wgrad_idx: torch.Tensor, n_params: int, splits: Optional[int] = None,
                   wgrad_precision: str = "bf16x3", data_precision: str = "fp32",
                   out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dL/d raw [N,S,C] + save record -> flat parameter gradient (state_dict order).  ``data_precision`` /
    ``wgrad_precision``: arithmetic of the backward-data chain and of the weight-gradient GEMMs: "fp32" (exact
    MFMA), "bf16x6" (six bf16 MFMAs per product on exact three-way splits of both operands: fp32-grade) or "bf16x3"
    (three bf16 MFMAs per product, 16 operand bits); ``bwd_blob`` must be the blob of ``data_precision``
    (``DM_NeRF.blob(BWD_BLOB[data_precision])``).
    ``out``: flat fp32 destination (the scatter table covers every parameter exactly once, so it is overwritten)."""
    for prec in (wgrad_precision, data_precision):
        if prec not in BACKWARD_PRECISIONS:
            raise ValueError(f"backward precisions are {BACKWARD_PRECISIONS} (gradients underflow fp16)")
    _dev(g_raw, "g_raw")
    g_raw = g_raw.contiguous()
    m = g_raw.shape[0] * g_raw.shape[1]
    if g_raw.shape[2] != 5 + ins_num:
        raise ValueError(f"g_raw must have {5 + ins_num} channels")
    lib = load()
    dev = g_raw.device
    if save.numel() != lib.dmn_field_save_floats(m):
        raise ValueError("save record does not belong to this many points")
    n_tiles = (m + 127) // 128 * 4
    if splits is None:
        splits = max(1, min(85, n_tiles // 8))          # 9 big GEMMs x 85 slices = 3 workgroups per CU
    grec = torch.empty(lib.dmn_field_grad_floats(m, ins_num), dtype=torch.float32, device=dev)
    with _timed("field_bwd_data", m):
        if data_precision == "fp32":
            check(lib.dmn_field_bwd_data(ptr(g_raw), ptr(save), ptr(bwd_blob), m, ins_num, ptr(grec),
                                         current_stream()), "dmn_field_bwd_data")
        elif data_precision == "bf16x6":
            check(lib.dmn_field_bwd_data_x6(ptr(g_raw), ptr(save), ptr(bwd_blob), m, ins_num, ptr(grec),
                                            current_stream()), "dmn_field_bwd_data_x6")
        else:
            check(lib.dmn_field_bwd_data_x3(ptr(g_raw), ptr(save), ptr(bwd_blob), m, ins_num, ptr(grec),
                                            current_stream()), "dmn_field_bwd_data_x3")
    work = torch.empty(lib.dmn_field_wgrad_workspace_floats(ins_num, splits), dtype=torch.float32, device=dev)
    if out is None:
        grads = torch.empty(n_params, dtype=torch.float32, device=dev)
    else:
        if out.dtype != torch.float32 or out.numel() != n_params or not out.is_contiguous() or out.device != dev:
            raise ValueError("out must be a contiguous float32 tensor with one entry per parameter on the same device")
        grads = out
    if m == 0:
        return grads.zero_()
    with _timed("field_bwd_weights", m):
        check(lib.dmn_field_bwd_weights(ptr(save), ptr(grec), m, ins_num, ptr(wgrad_idx), splits,
                                        precision_code(wgrad_precision), ptr(work), ptr(grads), 0,
                                        current_stream()), "dmn_field_bwd_weights")
    return grads


def field_embedded(embedded: torch.Tensor, blob: torch.Tensor, ins_num: int, precision: str = "fp32") -> torch.Tensor:
    _dev(embedded, "embedded")
    if embedded.shape[-1] != 90:
        raise ValueError(f"embedded input must have 63+27 channels, got {embedded.shape[-1]}")
    flat = embedded.reshape(-1, 90).contiguous()
    raw = torch.empty(flat.shape[0], 5 + ins_num, dtype=torch.float32, device=embedded.device)
    if precision == "fp32":
        check(load().dmn_field_fwd_embedded(ptr(flat), flat.shape[0], ptr(blob), ins_num, ptr(raw),
                                            current_stream()), "dmn_field_fwd_embedded")
    elif precision in X6_PRECISIONS:
        fn = load().dmn_field_fwd_embedded_x6w if uses_x6w(ins_num, precision) else load().dmn_field_fwd_embedded_x6
        check(fn(ptr(flat), flat.shape[0], ptr(blob), ins_num, ptr(raw), current_stream()), "dmn_field_fwd_embedded_x6")
    else:
        check(load().dmn_field_fwd_embedded_x3(ptr(flat), flat.shape[0], ptr(blob), ins_num,
                                               precision_code(precision), ptr(raw), ptr(status_word(raw.device)),
                                               current_stream()), "dmn_field_fwd_embedded_x3")
    return raw.reshape(*embedded.shape[:-1], raw.shape[-1])


def composite(raw: torch.Tensor, z: torch.Tensor, rays_d: torch.Tensor, keep_last: bool = False):
    _dev(raw, "raw")
    n, s, c = raw.shape
    raw = raw.contiguous()
    zz, stride = _rows(z, n, "z_vals")
    if z.shape[1] != s:
        raise ValueError("raw and z_vals disagree on the sample count")
    rays_d = _dev(rays_d, "rays_d").contiguous()
    k = c - 4 if keep_last else c - 5
    dev = raw.device
    rgb = torch.empty(n, 3, dtype=torch.float32, device=dev)
    weights = torch.empty(n, s, dtype=torch.float32, device=dev)
    depth = torch.empty(n, dtype=torch.float32, device=dev)
    ins = torch.empty(n, k, dtype=torch.float32, device=dev)
    with _timed("composite", n * s):
        check(load().dmn_composite_fwd(ptr(raw), ptr(zz), stride, ptr(rays_d), n, s, c, int(keep_last), ptr(rgb),
                                       ptr(weights), ptr(depth), ptr(ins), current_stream()), "dmn_composite_fwd")
    return rgb, weights, depth, ins


FUSED_MIN_CHANNELS = 32      # = kWideChannels of the compositing kernels (csrc/composite_common.h)
FUSED_SCRATCH_FLOATS = 2048  # = kFusedExtraBytes / 2 / 4 of csrc/field_fused_x3.hip (per ray of the workgroup's pair)


def fused_supported(n_samples: int, n_importance: int, ins_num: int, precision: str) -> bool:
    """Whether ``render_fused`` covers this configuration: bf16 / fp16 matrix-pipe kernels, 64-sample passes, and the coarse
    launch's resampling scratch -- (4 * n_samples + n_importance) floats per ray -- inside its 8 KB of LDS (the same
    bound ``dmn_field_composite_x3`` checks; beyond it the image loop takes the chained kernels)."""
    return (precision in X3_PRECISIONS + X6_PRECISIONS and n_samples % 64 == 0 and (n_samples + n_importance) % 64 == 0
            and n_samples >= 64 and 0 <= n_importance <= 1024 and 4 * n_samples + n_importance <= FUSED_SCRATCH_FLOATS
            and ins_num + 1 <= MAX_INS_CHANNELS)


def field_composite(rays_o: torch.Tensor, rays_d: torch.Tensor, z: torch.Tensor, blob: torch.Tensor, ins_num: int,
                    precision: str, sigma_only: bool = False, maps: bool = True, weights: bool = False,
                    n_new: Optional[int] = None, u: Optional[torch.Tensor] = None):
    """Field evaluation + compositing in one launch (``raw`` never reaches memory): returns
    ``(rgb, depth, ins, weights, z_fine)`` with ``None`` for what was not asked for.  ``n_new`` not None: the
    resampling + sorted merge of dm_nerf on the composited weights (``u``: draws, ``None`` = det) -> ``z_fine``."""
    n = rays_o.shape[0]
    rays_o = _dev(rays_o, "rays_o").contiguous()
    rays_d = _dev(rays_d, "rays_d").contiguous()
    zz, stride = _rows(z, n, "z_vals")
    s = z.shape[1]
    dev = rays_o.device
    new = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)
    rgb, depth, ins = (new(n, 3), new(n), new(n, ins_num)) if maps else (None, None, None)
    w = new(n, s) if weights else None
    z_fine, uu, ustride = None, None, 0
    if n_new is not None:
        uu, ustride = _draws(u, n, n_new)
        z_fine = new(n, s + n_new)
    with _timed("field_fwd", n * s):
        check(load().dmn_field_composite_x3(ptr(rays_o), ptr(rays_d), ptr(zz), stride, n, s, ptr(blob), ins_num,
                                            precision_code(precision), int(sigma_only), ptr(rgb), ptr(depth), ptr(ins),
                                            ptr(w), ptr(uu), ustride, n_new or 0, ptr(z_fine),
                                            ptr(status_word(dev)), current_stream()), "dmn_field_composite_x3")
    return rgb, depth, ins, w, z_fine


def render_fused(rays_o: torch.Tensor, rays_d: torch.Tensor, z_coarse: torch.Tensor, blob_coarse: torch.Tensor,
                 blob_fine: torch.Tensor, ins_num: int, precision: str, n_importance: int,
                 u: Optional[torch.Tensor] = None, coarse_maps: bool = False, coarse_sigma_only: bool = False):
    """One inference chunk of dm_nerf in two launches (coarse field + compositing + resampling, fine field +
    compositing).  Returns a dict with rgb_fine, depth_fine, ins_fine, z_vals_fine (+ rgb_coarse, depth_coarse,
    ins_coarse with ``coarse_maps``); bit-identical to the unfused chain."""
    rgb_c, depth_c, ins_c, _, z_fine = field_composite(rays_o, rays_d, z_coarse, blob_coarse, ins_num, precision,
                                                       sigma_only=coarse_sigma_only, maps=coarse_maps,
                                                       n_new=int(n_importance), u=u)
    rgb, depth, ins, _, _ = field_composite(rays_o, rays_d, z_fine, blob_fine, ins_num, precision)
    out = {"rgb_fine": rgb, "depth_fine": depth, "ins_fine": ins, "z_vals_fine": z_fine}
    if coarse_maps:
        out.update({"rgb_coarse": rgb_c, "depth_coarse": depth_c, "ins_coarse": ins_c})
    return out


def sample_pdf(bins: torch.Tensor, weights: torch.Tensor, n_new: int, u: Optional[torch.Tensor]) -> torch.Tensor:
    n, b = bins.shape
    bb, bstride = _rows(bins, n, "bins")
    w = _dev(weights, "weights")
    if w.shape != (n, b - 1):
        raise ValueError(f"weights must be [{n}, {b - 1}], got {tuple(w.shape)}")
    if w.stride(1) != 1:
        w = w.contiguous()
    wstride = w.stride(0) if n > 1 else b - 1
    u, ustride = _draws(u, n, n_new)
    out = torch.empty(n, n_new, dtype=torch.float32, device=bins.device)
    check(load().dmn_sample_pdf(ptr(bb), bstride, ptr(w), wstride, ptr(u), ustride, n, b, n_new, ptr(out),
                                current_stream()), "dmn_sample_pdf")
    return out


def resample_merge(z: torch.Tensor, weights: torch.Tensor, n_new: int, u: Optional[torch.Tensor]) -> torch.Tensor:
    n, s = z.shape
    zz, stride = _rows(z, n, "z_vals")
    weights = _dev(weights, "weights").contiguous()
    u, ustride = _draws(u, n, n_new)
    out = torch.empty(n, s + n_new, dtype=torch.float32, device=z.device)
    check(load().dmn_resample_merge(ptr(zz), stride, ptr(weights), ptr(u), ustride, n, s, n_new, ptr(out),
                                    current_stream()), "dmn_resample_merge")
    return out


def merge_depths(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    n = a.shape[0]
    aa, astride = _rows(a, n, "a")
    bb, bstride = _rows(b, n, "b")
    out = torch.empty(n, a.shape[1] + b.shape[1], dtype=torch.float32, device=a.device)
    check(load().dmn_merge_depths(ptr(aa), astride, a.shape[1], ptr(bb), bstride, b.shape[1], n, ptr(out),
                                  current_stream()), "dmn_merge_depths")
    return out
