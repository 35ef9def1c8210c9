"""Timing probe (GPU box): the field kernel of every arithmetic on the bench's launch shapes (65536-ray chunks, S = 64 and
192), HIP events around each launch, interleaved rounds on one device.  Prints ms per launch and algorithmic TFLOP/s."""
import os
import sys
import time

import numpy as np
import torch

import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dm_nerf_amd as dm
from dm_nerf_amd import ops, synthetic

INS = int(os.environ.get("PROBE_INS", "13"))
FLOP_PER_POINT = 2 * 693504


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    modes = sys.argv[2].split(",") if len(sys.argv) > 2 else ["fp32", "bf16x3", "bf16x6"]
    dev = torch.device("cuda:0")
    model = dm.DM_NeRF(8, 256, 63, 27, [4], INS).to(dev)
    model.load_state_dict(synthetic.make_params(INS, 0))
    n = 65536
    rng = np.random.default_rng(0)
    o = torch.from_numpy(rng.uniform(-1, 1, (n, 3)).astype(np.float32)).to(dev)
    d = torch.from_numpy(rng.standard_normal((n, 3)).astype(np.float32)).to(dev)
    res = {}
    for s in (64, 192):
        z = torch.sort(torch.rand(n, s, device=dev) * 11 + 4, -1)[0]
        for m in modes:
            ops.field_rays(o, d, z, model.blob(m), INS, m)
        torch.cuda.synchronize()
        for r in range(rounds):
            for m in modes:
                t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
                t0.record()
                raw = ops.field_rays(o, d, z, model.blob(m), INS, m)
                t1.record()
                torch.cuda.synchronize()
                res.setdefault((s, m), []).append(t0.elapsed_time(t1))
                del raw
    for (s, m), v in res.items():
        ms = float(np.median(v))
        print(f"S={s:3d} {m:7s} {ms:8.3f} ms  {n * s * FLOP_PER_POINT / ms / 1e9:8.1f} TFLOP/s   all: " + " ".join(f"{x:.2f}" for x in v), flush=True)
    for m in modes:
        tot = float(np.median(res[(64, m)])) + float(np.median(res[(192, m)]))
        print(f"{m}: coarse + fine per 65536 rays {tot:.2f} ms -> {n / tot / 1e-3 / 1e3:.0f} k rays/s (field kernels only)")


if __name__ == "__main__":
    main()
