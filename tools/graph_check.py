import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
import dm_nerf_amd as dm
from dm_nerf_amd import trainer, synthetic
dev = torch.device('cuda:0')
n, ins = 4096, 13
def setup():
    args = dm.default_args(ins_num=ins, perturb=1.0, is_train=True, device=dev)
    pos, view, coarse, fine, args = dm.create_nerf(args)
    coarse.load_state_dict(synthetic.make_params(ins, 0)); fine.load_state_dict(synthetic.make_params(ins, 1))
    coarse.train(); fine.train()
    args.tolerance, args.deta_w = 0.05, 0.05
    K, c2w = synthetic.dmsr_intrinsics(800, 800), synthetic.study_pose(1)
    ro, rd = dm.get_rays_k(800, 800, K, c2w)
    g = torch.Generator(device=dev).manual_seed(7)
    pick = torch.randperm(640000, device=dev, generator=g)[:n]
    rays = torch.stack([ro.reshape(-1, 3)[pick], rd.reshape(-1, 3)[pick]], 0)
    target = torch.rand(n, 3, device=dev, generator=g)
    labels = torch.randint(0, ins, (n,), device=dev, generator=g)
    z = dm.z_val_sample(n, args.near, args.far, args.N_samples)
    return (pos, view, coarse, fine, args), rays, target, labels, z
nerf, rays, target, labels, z = setup()
pen = lambda raw, zz, depth: dm.ins_penalizer(raw, zz, depth, rays[1], nerf[4])
step = trainer.GraphedTrainStep(rays, target, nerf, z, ins_loss=trainer.instance_loss(labels, ins, assign="device"), penalizer=pen)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(20): terms = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 20
print("graphed step", dt * 1e3, "ms", {k: float(v) for k, v in terms.items()}, "lr", float(step.lr), "it", float(step.iteration))
# eager for comparison
nerf2, rays2, target2, labels2, z2 = setup()
opt = trainer.make_optimizer(nerf2[2], nerf2[3])
pen2 = lambda raw, zz, depth: dm.ins_penalizer(raw, zz, depth, rays2[1], nerf2[4])
il = trainer.instance_loss(labels2, ins, assign="device")
for i in range(3): trainer.train_step(i, rays2, target2, nerf2, z2, opt, ins_loss=il, penalizer=pen2, report=False)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(20): last = trainer.train_step(3 + i, rays2, target2, nerf2, z2, opt, ins_loss=il, penalizer=pen2, report=False)
torch.cuda.synchronize()
print("eager step", (time.perf_counter() - t0) / 20 * 1e3, "ms", {k: float(v) for k, v in last.items()})
