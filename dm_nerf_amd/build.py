"""Builds ``csrc/libdmnerf_hip.so`` (gfx950 code objects + C ABI) with hipcc, in-tree.

    python -m dm_nerf_amd.build [--force]

hipcc cross-compiles for gfx950 without a GPU, so this runs in the build container; the
resulting ``.so`` travels to the GPU box with the repo snapshot (it is git-ignored).
Every source is compiled to its own object (in parallel, only when it or a header changed),
then linked: a one-kernel edit rebuilds in the time of that kernel."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libdmnerf_hip.so")
OBJ = os.path.join(CSRC, "build")
SOURCES = ["field_fwd.hip", "field_fwd_x3.hip", "field_fwd_x6.hip", "field_fwd_x6w.hip", "field_fused_x3.hip", "field_bwd.hip", "field_bwd_x3.hip", "field_bwd_x6.hip", "render_ops.hip",
           "composite_bwd.hip", "penalizer.hip", "ins_stats.hip", "exchange.hip", "field_pack.cpp", "render_chain.cpp"]
HEADERS = ["field_layout.h", "field_common.h", "field_stream.h", "field_x3.h", "field_x3_net.h", "field_x6.h", "composite_common.h", "trig.h",
           os.path.join("..", "..", "include", "dmnerf_hip.h")]
# -ffp-contract=off: the path reproduces separately rounded fp32 multiplies and adds of the
# reference (e.g. pts = o + d*z feeds a 2^9 frequency band); fused multiply-adds are explicit.
# -pragma-unroll-threshold: the field kernels are fully unrolled on purpose (register arrays need
# compile-time indices); the default 16k-instruction cap would silently leave loops rolled -> scratch.
FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-std=c++17",
         "-mllvm", "-pragma-unroll-threshold=400000"]


def _existing(names):
    return [f for f in names if os.path.exists(os.path.join(CSRC, f))]


def _newest_header() -> float:
    return max(os.path.getmtime(os.path.join(CSRC, h)) for h in _existing(HEADERS))


def _obj(src: str) -> str:
    return os.path.join(OBJ, src + ".o")


def _stale_objects(force: bool):
    hdr = _newest_header()
    out = []
    for src in _existing(SOURCES):
        o = _obj(src)
        if force or not os.path.exists(o) or os.path.getmtime(o) < max(hdr, os.path.getmtime(os.path.join(CSRC, src))):
            out.append(src)
    return out


def build(force: bool = False, verbose: bool = False, extra_flags=(), lib: str = LIB, obj_dir: str = None) -> str:
    """``extra_flags`` / ``lib`` / ``obj_dir``: variant builds of the same ABI for same-device A/B timing
    (``tools/ab.sh``; ``DMNERF_LIB`` selects the library at run time)."""
    global OBJ
    saved = OBJ
    if obj_dir is not None:
        OBJ = obj_dir
    try:
        sources = _existing(SOURCES)
        todo = _stale_objects(force)
        if not todo and os.path.exists(lib) and all(os.path.getmtime(lib) >= os.path.getmtime(_obj(s)) for s in sources):
            return lib
        hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
        if not os.path.exists(hipcc):
            raise RuntimeError("hipcc not found: cannot build the gfx950 extension")
        os.makedirs(OBJ, exist_ok=True)

        def compile_one(src):
            cmd = [hipcc, *FLAGS, *extra_flags, "-c", src, "-o", _obj(src)]
            if verbose:
                print(" ".join(cmd), flush=True)
            proc = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True)
            if proc.returncode != 0:
                raise RuntimeError(f"hipcc failed on {src} ({proc.returncode}):\n{proc.stdout}\n{proc.stderr}")

        with ThreadPoolExecutor(max_workers=min(6, max(1, len(todo)))) as pool:
            list(pool.map(compile_one, todo))
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *[_obj(s) for s in sources], "-o", lib + ".tmp"]
        proc = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True)
        if proc.returncode != 0:
            raise RuntimeError(f"link failed ({proc.returncode}):\n{proc.stdout}\n{proc.stderr}")
        os.replace(lib + ".tmp", lib)
        return lib
    finally:
        OBJ = saved


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
