"""Builds ``csrc/libdmnerf_hip.so`` (gfx950 code objects + C ABI) with hipcc, in-tree.

    python -m dm_nerf_amd.build [--force]

hipcc cross-compiles for gfx950 without a GPU, so this runs in the build container; the
resulting ``.so`` travels to the GPU box with the repo snapshot (it is git-ignored)."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libdmnerf_hip.so")
SOURCES = ["field_fwd.hip", "field_fwd_x3.hip", "field_bwd.hip", "field_bwd_x3.hip", "render_ops.hip", "composite_bwd.hip", "penalizer.hip", "ins_stats.hip", "exchange.hip",
           "field_pack.cpp", "render_fused.cpp"]
HEADERS = ["field_layout.h", "field_common.h", "field_stream.h", "field_x3.h", "trig.h", os.path.join("..", "..", "include", "dmnerf_hip.h")]
# -ffp-contract=off: the path reproduces separately rounded fp32 multiplies and adds of the
# reference (e.g. pts = o + d*z feeds a 2^9 frequency band); fused multiply-adds are explicit.
# -pragma-unroll-threshold: the field kernels are fully unrolled on purpose (register arrays need
# compile-time indices); the default 16k-instruction cap would silently leave loops rolled -> scratch.
FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17",
         "-mllvm", "-pragma-unroll-threshold=400000"]


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    built = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > built for f in SOURCES + HEADERS)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not _stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build the gfx950 extension")
    cmd = [hipcc, *FLAGS, *SOURCES, "-o", LIB + ".tmp"]
    if verbose:
        print(" ".join(cmd), flush=True)
    proc = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError(f"hipcc failed ({proc.returncode}):\n{proc.stdout}\n{proc.stderr}")
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
