"""Synthetic workload of the benchmarks and tests: seeded weights in the reference's ``state_dict`` layout and the
DM-SR 'study' cameras (no dataset or checkpoint is available offline).  Data generators only --
nothing of the render path is computed here.  The test-side CPU checker re-exports these so that both sides of a
parity test build the same weights from the same seed."""
from __future__ import annotations

import math
from typing import Dict, List, Tuple

import numpy as np
import torch

NET_DEPTH, NET_WIDTH, SKIP_AFTER = 8, 256, 4          # config.py:23,25,153
POS_CH, DIR_CH = 63, 27                               # 3 + 6*10, 3 + 6*4 (config.py:56,58)


def param_specs(ins_num: int, width: int = NET_WIDTH, depth: int = NET_DEPTH) -> List[Tuple[str, Tuple[int, ...]]]:
    """Names and shapes of one field network, in the reference's ``state_dict`` order
    (registration order of ``DM_NeRF.__init__``, networks/dm_nerf.py:59-78)."""
    specs: List[Tuple[str, Tuple[int, ...]]] = []
    for i in range(depth):
        fan_in = POS_CH if i == 0 else (width + POS_CH if i == SKIP_AFTER + 1 else width)
        specs.append((f"mlps.{i}.weight", (width, fan_in)))
        specs.append((f"mlps.{i}.bias", (width,)))
    specs += [
        ("rgb_feature_linear.weight", (width, width)),
        ("rgb_feature_linear.bias", (width,)),
        ("ins_feature_linear.weight", (width, width)),
        ("ins_feature_linear.bias", (width,)),
        ("rgb_feature_linears.0.weight", (width // 2, width + DIR_CH)),
        ("rgb_feature_linears.0.bias", (width // 2,)),
        ("ins_feature_linears.0.weight", (width // 2, width)),
        ("ins_feature_linears.0.bias", (width // 2,)),
        ("density_linear.weight", (1, width)),
        ("density_linear.bias", (1,)),
        ("ins_linear.weight", (ins_num + 1, width // 2)),
        ("ins_linear.bias", (ins_num + 1,)),
        ("rgb_linear.weight", (3, width // 2)),
        ("rgb_linear.bias", (3,)),
    ]
    return specs


def make_params(ins_num: int, seed: int, sharpen: bool = True) -> Dict[str, torch.Tensor]:
    """Deterministic synthetic weights: PCG64(seed), U(+-1/sqrt(fan_in)) for weight and bias (the distribution of
    ``nn.Linear``'s default init).  ``sharpen`` scales the density head so that random fields contain opaque
    structure."""
    rng = np.random.default_rng(seed)
    out: Dict[str, torch.Tensor] = {}
    fan_in = 1
    for name, shape in param_specs(ins_num):
        if name.endswith(".weight"):
            fan_in = shape[1]
        bound = 1.0 / math.sqrt(fan_in)
        out[name] = torch.from_numpy(rng.uniform(-bound, bound, size=shape).astype(np.float32))
    if sharpen:
        out["density_linear.weight"] = out["density_linear.weight"] * 30.0
        out["density_linear.bias"] = out["density_linear.bias"] + 0.5
    return out


def dmsr_intrinsics(H: int, W: int, camera_angle_x: float = 0.6911) -> np.ndarray:
    """``[[f,0,W/2],[0,-f,H/2],[0,0,-1]]`` (datasets/loader_dmsr.py:171-172)."""
    f = 0.5 * W / math.tan(0.5 * camera_angle_x)
    return np.array([[f, 0.0, W / 2.0], [0.0, -f, H / 2.0], [0.0, 0.0, -1.0]], dtype=np.float64)


def pose_spherical(theta_deg: float, phi_deg: float, radius: float) -> np.ndarray:
    """Blender-style orbit pose (datasets/loader_dmsr.py:10-34)."""
    t = np.eye(4); t[2, 3] = radius
    p = math.radians(phi_deg)
    rp = np.array([[1, 0, 0, 0], [0, math.cos(p), -math.sin(p), 0], [0, math.sin(p), math.cos(p), 0], [0, 0, 0, 1.0]])
    th = math.radians(theta_deg)
    rt = np.array([[math.cos(th), 0, -math.sin(th), 0], [0, 1, 0, 0], [math.sin(th), 0, math.cos(th), 0], [0, 0, 0, 1.0]])
    flip = np.array([[-1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1.0]])
    return (flip @ rt @ rp @ t).astype(np.float32)


def study_pose(index: int = 0) -> np.ndarray:
    """Pose 0 = identity rotation at (0.3,-0.2,6.0); poses 1..7 orbit at phi=-65deg, r=7."""
    if index == 0:
        c2w = np.eye(4, dtype=np.float32)
        c2w[:3, 3] = (0.3, -0.2, 6.0)
        return c2w
    return pose_spherical(-180.0 + 45.0 * index, -65.0, 7.0)
