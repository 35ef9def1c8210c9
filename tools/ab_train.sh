#!/usr/bin/env bash
# Same-device A/B of library variants on the training step (tools/bench_train.py), interleaved rounds.
#   bash tools/ab_train.sh lib1.so lib2.so ...
set -euo pipefail
cd "$(dirname "$0")/.."
for rep in 1 2 3; do
  for lib in "$@"; do
    DMNERF_LIB=$PWD/$lib python - <<'PY'
import os, sys, json, time, torch
sys.path.insert(0, os.getcwd())
import dm_nerf_amd as dm
from dm_nerf_amd import ops, synthetic, trainer
dev = torch.device("cuda", 0)
args = dm.default_args(ins_num=13, perturb=1.0, is_train=True, precision=os.environ.get("AB_PREC", "bf16x3"), device=dev)
pos, view, coarse, fine, args = dm.create_nerf(args)
coarse.load_state_dict(synthetic.make_params(13, 0)); fine.load_state_dict(synthetic.make_params(13, 1))
K, c2w = synthetic.dmsr_intrinsics(800, 800), synthetic.study_pose(1)
ro, rd = dm.get_rays_k(800, 800, K, c2w)
g = torch.Generator(device=dev).manual_seed(7)
pick = torch.randperm(640000, device=dev, generator=g)[:4096]
target = torch.rand(4096, 3, device=dev, generator=g)
rays = torch.stack([ro.reshape(-1, 3)[pick], rd.reshape(-1, 3)[pick]], 0)
z = dm.z_val_sample(4096, args.near, args.far, args.N_samples)
opt = trainer.make_optimizer(coarse, fine)
step = lambda i: trainer.train_step(i, rays, target, (pos, view, coarse, fine, args), z, opt)
step(0); torch.cuda.synchronize()
ops.PROFILE = []
t0 = time.perf_counter()
for i in range(6): step(i + 1)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 6
prof = ops.PROFILE
ms = lambda name: sum(a.elapsed_time(b) for k, a, b, w in prof if k == name) / 6
print(os.path.basename(os.environ["DMNERF_LIB"]), f"step {dt*1e3:.2f} ms  fwd_save {ms('field_fwd_save'):.2f}  bwd_data {ms('field_bwd_data'):.2f}  wgrad {ms('field_bwd_weights'):.2f}")
PY
  done
done
