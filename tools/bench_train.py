#!/usr/bin/env python3
"""Training-step timing (BASELINE config 4 shape on one GPU): N_rand rays per step, stochastic path,
colour losses, Adam.  Prints one JSON line.  Not the headline bench (bench.py is)."""
import argparse, json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dm_nerf_amd as dm
from dm_nerf_amd import trainer
from dm_nerf_amd import synthetic as oracle   # seeded synthetic weights and cameras

ap = argparse.ArgumentParser()
ap.add_argument("--rays", type=int, default=4096)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--precision", default="fp32")
ap.add_argument("--instance", action="store_true", help="add the instance-matching loss on random labels")
ap.add_argument("--penalize", action="store_true", help="add the emptiness penalizer (configs: tolerance = deta_w = 0.05)")
ns = ap.parse_args()
dev = torch.device("cuda", 0)
args = dm.default_args(ins_num=13, perturb=1.0, is_train=True, precision=ns.precision, device=dev)
pos, view, coarse, fine, args = dm.create_nerf(args)
coarse.load_state_dict(oracle.make_params(13, 0)); fine.load_state_dict(oracle.make_params(13, 1))
K, c2w = oracle.dmsr_intrinsics(800, 800), oracle.study_pose(1)
ro, rd = dm.get_rays_k(800, 800, K, c2w)
pick = torch.randperm(640000, device=dev)[: ns.rays]
rays = torch.stack([ro.reshape(-1, 3)[pick], rd.reshape(-1, 3)[pick]], 0)
target = torch.rand(ns.rays, 3, device=dev)
z = dm.z_val_sample(ns.rays, args.near, args.far, args.N_samples)
opt = trainer.make_optimizer(coarse, fine)
args.tolerance, args.deta_w = 0.05, 0.05
labels = torch.randint(0, 13, (ns.rays,), device=dev)
ins = (lambda pred: dm.ins_criterion(pred, labels, 13)[0]) if ns.instance else None
pen = (lambda raw, zz, depth: dm.ins_penalizer(raw, zz, depth, rays[1], args)) if ns.penalize else None
trainer.train_step(0, rays, target, (pos, view, coarse, fine, args), z, opt, ins_loss=ins, penalizer=pen)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(ns.steps):
    trainer.train_step(i + 1, rays, target, (pos, view, coarse, fine, args), z, opt, ins_loss=ins, penalizer=pen)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / ns.steps
print(json.dumps({"metric": "training rays/s (HIP field fwd+save, HIP composite bwd, HIP field backward-data + weight-gradient kernels, Adam)",
                  "rays_per_step": ns.rays, "ms_per_step": dt * 1e3, "value": ns.rays / dt, "unit": "rays/s",
                  "precision_forward": ns.precision, "emptiness_penalizer": bool(ns.penalize), "instance_loss": bool(ns.instance), "peak_mem_gb": torch.cuda.max_memory_allocated() / 2**30}))
