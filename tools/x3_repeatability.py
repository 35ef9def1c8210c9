#!/usr/bin/env python3
"""Repeatability probe of the split-precision field kernels: several shapes, three launches each, error against the
exact kernel.  Differences between repetitions mean a race or an uncovered hardware hazard (this is how the
inline-asm accumulator-read hazard showed up: 7e-2 errors that changed from run to run)."""
import sys, torch
sys.path.insert(0, '/root/repo')
import dm_nerf_amd as dm
from dm_nerf_amd import ops
from dm_nerf_amd import synthetic as oracle
dev = torch.device('cuda', 0)
m = dm.DM_NeRF(8, 256, 63, 27, [4], 13).to(dev)
m.load_state_dict(oracle.make_params(13, 1))
torch.manual_seed(0)
for (n, s) in ((100, 64), (100, 192), (80, 64), (128, 64), (2, 64), (37, 64), (4096, 64), (1, 192)):
    o = torch.rand(n, 3, device=dev); d = torch.randn(n, 3, device=dev)
    z = torch.sort(torch.rand(n, s, device=dev) * 11 + 4, -1).values
    ref = ops.field_rays(o, d, z, m.blob('fp32'), 13, 'fp32')
    for prec in ('fp16x3', 'bf16x3'):
        for rep in range(3):
            got = ops.field_rays(o, d, z, m.blob(prec), 13, prec)
            torch.cuda.synchronize()
            err = float((got - ref).abs().max() / ref.abs().max())
            bad = ((got - ref).abs().amax(-1) > 1e-4 * ref.abs().max()).float().mean().item()
            print(n, s, prec, rep, f'{err:.2e}', f'bad points {bad:.3f}', 'status', ops.take_status(dev))
