import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
import dm_nerf_amd as dm
from dm_nerf_amd import ops, synthetic
dev = torch.device('cuda:0')
for ins in (13, 40):
    model = dm.DM_NeRF(8, 256, 63, 27, [4], ins).to(dev)
    model.load_state_dict(synthetic.make_params(ins, 1))
    rng = np.random.default_rng(0)
    for n, s in ((37, 64), (5, 192), (1, 7)):
        o = torch.from_numpy(rng.uniform(-1, 1, (n, 3)).astype(np.float32)).to(dev)
        d = torch.from_numpy(rng.standard_normal((n, 3)).astype(np.float32)).to(dev)
        z = torch.sort(torch.rand(n, s, device=dev) * 11 + 4, -1)[0]
        ref = ops.field_rays(o, d, z, model.blob('fp32'), ins, 'fp32')
        a = ops.field_rays(o, d, z, model.blob('bf16x6_32'), ins, 'bf16x6_32')
        b = ops.field_rays(o, d, z, model.blob('bf16x6'), ins, 'bf16x6')
        sc = float(ref.abs().max())
        print(ins, n, s, 'x6_32 vs fp32', float((a - ref).abs().max()) / sc, ' x6w vs fp32', float((b - ref).abs().max()) / sc, ' finite', bool(torch.isfinite(b).all()))
        bs = ops.field_rays(o, d, z, model.blob('bf16x6'), ins, 'bf16x6', sigma_only=True)
        print('   sigma_only: sigma equal', bool(torch.equal(bs[..., 3], b[..., 3])), 'rest zero', float(bs[..., :3].abs().max()) == 0 and float(bs[..., 4:].abs().max()) == 0)
