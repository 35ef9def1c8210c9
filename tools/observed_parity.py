#!/usr/bin/env python3
"""Prints what the end-to-end parity tests observe per precision, so that their caps can be set to ~2x the observation
(tests/test_gpu_parity.py::_PATH_CAPS) instead of a blanket bound.  GPU box:  python tools/observed_parity.py"""
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import cases  # noqa: E402
import dm_nerf_amd as dm  # noqa: E402
from oracle import dmnerf_cpu as oracle  # noqa: E402

DEV = "cuda"
G = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
Hn = lambda t: t.detach().cpu().numpy()


def stats(got, want):
    per_ray = np.abs(got - want).reshape(got.shape[0], -1).max(-1) / max(np.abs(want).max(), 1e-30)
    return f"off {np.mean(per_ray > 1e-4):.4f} max {per_ray.max():.2e}"


def nets(ins_num, sc, sf):
    args = dm.default_args(ins_num=ins_num)
    pos, view, coarse, fine, args = dm.create_nerf(args)
    coarse.load_state_dict(oracle.make_params(ins_num, sc)); fine.load_state_dict(oracle.make_params(ins_num, sf))
    return pos, view, coarse.eval(), fine.eval(), args


g6 = np.load(os.path.join(ROOT, "tests/golden/g6_path.npz"))
g8 = np.load(os.path.join(ROOT, "tests/golden/g8_image.npz"))
n = 128
o, d, t_rand, u = cases.path_rays(n)
for prec in ("fp32", "fp16x3", "bf16x3"):
    pos, view, coarse, fine, args = nets(13, 0, 1)
    args.precision = prec
    rays = G(np.stack([o, d], 0))
    z = dm.z_val_sample(n, cases.NEAR, cases.FAR, 64)
    with torch.no_grad():
        det = dm.dm_nerf(rays, pos, view, coarse, fine, z, args)
        a_r = SimpleNamespace(**{**vars(args), "perturb": 1.0, "is_train": True, "t_rand": G(t_rand), "u": G(u)})
        rnd = dm.dm_nerf(rays, pos, view, coarse, fine, z, a_r)
    for tag, out in (("det", det), ("rnd", rnd)):
        line = [f"{prec} G6 {tag}:"]
        for k in ("rgb_coarse", "ins_coarse", "depth_coarse", "rgb_fine", "ins_fine", "depth_fine"):
            line.append(f"{k} {stats(Hn(out[k]), g6[f'{tag}_{k}'])};")
        zf = Hn(out["z_vals_fine"]); zw = g6[f"{tag}_z_vals_fine"]
        line.append(f"z_fine frac>1e-4 {np.mean(np.abs(zf - zw) > 1e-4 * np.abs(zw).max()):.5f} max {np.abs(zf - zw).max() / np.abs(zw).max():.2e}")
        print(" ".join(line))
    args.N_test = 100
    Hh, W = 20, 24
    rgbs, inss, psnrs = dm.render_test(pos, view, coarse, fine, [cases.study_c2w()], [Hh, W, cases.study_intrinsics(Hh, W)],
                                       args, gt_imgs=[g8["rgb"]])
    print(f"{prec} G8: rgb {stats(Hn(rgbs[0]).reshape(-1, 3), g8['rgb'].reshape(-1, 3))}; ins "
          f"{stats(Hn(inss[0]).reshape(-1, 13), g8['ins'].reshape(-1, 13))}; psnr {psnrs[0]:.1f}")
    # full-size chunk sample vs oracle (test_full_size_chunk_properties)
    Hh = W = 800
    K, c2w = oracle.dmsr_intrinsics(Hh, W), oracle.study_pose(0)
    ro, rd = dm.get_rays_k(Hh, W, K, c2w)
    ro, rd = ro.reshape(-1, 3), rd.reshape(-1, 3)
    start, nn = 317 * 800 + 123, 4096
    rays = torch.stack([ro[start:start + nn], rd[start:start + nn]], 0)
    z = dm.z_val_sample(nn, args.near, args.far, 64)
    with torch.no_grad():
        full = dm.dm_nerf(rays, pos, view, coarse, fine, z, args)
    pick = np.random.default_rng(9).choice(nn, size=48, replace=False)
    with torch.no_grad():
        want = oracle.render_rays(rays[:, pick].cpu(), oracle.make_params(13, 0), oracle.make_params(13, 1),
                                  oracle.coarse_depths(48, args.near, args.far, 64))
    print(f"{prec} chunk sample:", "; ".join(f"{k} {stats(Hn(full[k])[pick], want[k].numpy())}" for k in ("rgb_fine", "ins_fine", "depth_fine")))
