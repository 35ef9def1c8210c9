"""field_fwd_x6w_kernel run repeatedly on the same inputs, and the same rays in two batch layouts: bit equality."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dm_nerf_amd as dm
from dm_nerf_amd import ops, synthetic
dev = torch.device('cuda:0')
ins = 13
model = dm.DM_NeRF(8, 256, 63, 27, [4], ins).to(dev)
model.load_state_dict(synthetic.make_params(ins, 1))
rng = np.random.default_rng(0)
n, s = 4096, 192
o = torch.from_numpy(rng.uniform(-1, 1, (n, 3)).astype(np.float32)).to(dev)
d = torch.from_numpy(rng.standard_normal((n, 3)).astype(np.float32)).to(dev)
z = torch.sort(torch.rand(n, s, device=dev) * 11 + 4, -1)[0]
for prec in ("bf16x6", "bf16x6_32"):
    blob = model.blob(prec)
    a = ops.field_rays(o, d, z, blob, ins, prec)
    bad = 0
    for rep in range(5):
        b = ops.field_rays(o, d, z, blob, ins, prec)
        bad += int((a != b).any(-1).sum())
    h = ops.field_rays(o[:1000], d[:1000], z[:1000], blob, ins, prec)
    print(prec, "points differing between repeated launches:", bad, " between batch layouts:", int((a[:1000] != h).any(-1).sum()),
          "of", 1000 * s)
