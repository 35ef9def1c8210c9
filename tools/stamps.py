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
from oracle import dmnerf_cpu as oracle  # noqa: E402  (seeded weights only)

dev = torch.device("cuda", 0)
m = dm.DM_NeRF(8, 256, 63, 27, [4], 13).to(dev)
m.load_state_dict(oracle.make_params(13, 1))
n, s = 4096, 192
o = torch.rand(n, 3, device=dev)
d = torch.randn(n, 3, device=dev)
z = torch.sort(torch.rand(n, s, device=dev) * 11 + 4, -1).values
ops._status[0] = torch.zeros(16, dtype=torch.int32, device=dev)   # 64-bit counters live behind the status word
for prec in ("fp16x3", "bf16x3"):
    blob = m.blob(prec)
    ops.field_rays(o, d, z, blob, 13, prec)
    torch.cuda.synchronize()
    ops._status[0].zero_()
    ops.field_rays(o, d, z, blob, 13, prec)
    torch.cuda.synchronize()
    c = ops._status[0].view(torch.int64).tolist()
    tot, dma, bar, cnt = c[1], c[2], c[3], max(c[4], 1)
    print(f"{prec}: waves sampled {cnt}; cycles per pass {tot / cnt:.0f}; waiting for own DMA at the barrier "
          f"{dma / cnt:.0f} ({100 * dma / max(tot, 1):.1f} %); waiting for the other waves {bar / cnt:.0f} "
          f"({100 * bar / max(tot, 1):.1f} %); MFMA-bound pass = 130560")
