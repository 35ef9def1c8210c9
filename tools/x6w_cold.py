"""First launches in a fresh process against later ones (cold LDS / caches): dm_nerf in bf16x6 twice, compare every output."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import cases
import dm_nerf_amd as dm
from dm_nerf_amd import ops, synthetic
dev = torch.device('cuda:0')
n, ins = 150, 13
args = dm.default_args(ins_num=ins)
pos, view, coarse, fine, args = dm.create_nerf(args)
coarse.load_state_dict(synthetic.make_params(ins, 0)); fine.load_state_dict(synthetic.make_params(ins, 1))
o, d, _, _ = cases.path_rays(n, seed=41)
rays = torch.from_numpy(np.stack([o, d], 0)).to(dev)
z = dm.z_val_sample(n, 4.0, 15.0, 64)
outs = []
with torch.no_grad():
    for rep in range(4):
        out = dm.dm_nerf(rays, pos, view, coarse, fine, z, args)
        torch.cuda.synchronize()
        outs.append({k: v.clone() for k, v in out.items()})
for rep in range(1, 4):
    print("run", rep, "vs run 0:", {k: int((outs[rep][k] != outs[0][k]).sum()) for k in outs[0]})
print("nan in run 0:", {k: int(torch.isnan(v).sum()) for k, v in outs[0].items()})
