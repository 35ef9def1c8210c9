#!/usr/bin/env python3
"""Calibration of FETCH_SIZE for composite_wide_kernel's access pattern (one dword per lane, rows of C floats):
the same launch at C = 96 (384-B rows = three whole 128-B lines: nothing to over-fetch) and at C = 98 (392-B rows, the
Replica head).  Run under rocprofv3 --pmc FETCH_SIZE; algorithmic read bytes are printed for comparison."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dm_nerf_amd import ops  # noqa: E402

n, s = 32768, 192
for c in (96, 98, 96, 98):
    raw = torch.randn(n, s, c, device="cuda")
    z = torch.sort(torch.rand(n, s, device="cuda") * 6.0, -1).values
    d = torch.randn(n, 3, device="cuda")
    junk = torch.empty(1 << 28, device="cuda").normal_()          # 1 GiB written: raw is out of the caches
    torch.cuda.synchronize()
    ops.composite(raw, z, d)
    torch.cuda.synchronize()
    print(f"C={c}: algorithmic read bytes {n * (s * c * 4 + s * 4 + 12)}")
    del raw, junk
