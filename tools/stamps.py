#!/usr/bin/env python3
"""Diagnostic (DMN_STAMPS build only): where a wave of the x3 field kernel spends its cycles.

    hipcc ... -DDMN_STAMPS ... -o dm_nerf_amd/csrc/abl/lib_stamps.so
    DMNERF_LIB=$PWD/dm_nerf_amd/csrc/abl/lib_stamps.so python tools/stamps.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dm_nerf_amd as dm  # noqa: E402
from dm_nerf_amd import ops  # noqa: E402
from dm_nerf_amd import synthetic as oracle  # noqa: E402  (seeded weights)

dev = torch.device("cuda", 0)
m = dm.DM_NeRF(8, 256, 63, 27, [4], 13).to(dev)
m.load_state_dict(oracle.make_params(13, 1))
ops._status[0] = torch.zeros(16, dtype=torch.int32, device=dev)   # 64-bit counters live behind the status word
for n, s in ((4096, 192), (32, 128)):        # 6144 workgroups (24 per CU) / 32 workgroups (an eighth of the CUs busy)
    o = torch.rand(n, 3, device=dev)
    d = torch.randn(n, 3, device=dev)
    z = torch.sort(torch.rand(n, s, device=dev) * 11 + 4, -1).values
    for prec in ("fp16x3", "bf16x3"):
        blob = m.blob(prec)
        for _ in range(3):
            ops.field_rays(o, d, z, blob, 13, prec)
        torch.cuda.synchronize()
        ops._status[0].zero_()
        ops.field_rays(o, d, z, blob, 13, prec)
        torch.cuda.synchronize()
        c = ops._status[0].view(torch.int64).tolist()
        tot, dma, bar, cnt, real = c[1], c[2], c[3], max(c[4], 1), max(c[5], 1)
        print(f"{n}x{s} {prec}: s_memtime runs at {tot / real * 100:.0f} MHz; workgroup wall time {real / cnt * 10:.0f} ns; "
              f"waves sampled {cnt}; cycles per workgroup {tot / cnt:.0f} (MFMA-bound 130560); waiting for own DMA at "
              f"the barrier {100 * dma / max(tot, 1):.1f} %; for the other waves {100 * bar / max(tot, 1):.1f} %")
