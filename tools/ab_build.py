#!/usr/bin/env python3
"""Variant builds of libdmnerf_hip.so for same-device A/B timing (tools/ab.sh): one library per -D flag set, objects
under tools/ab/obj/<name>/ (not shipped), libraries under tools/ab/<name>.so (git-ignored; they travel to the GPU box).

    python tools/ab_build.py base= scalar_sub=-DDMN_X3_SCALAR_SUB=1 ..."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dm_nerf_amd import build  # noqa: E402

out = os.path.join(ROOT, "tools", "ab")
os.makedirs(os.path.join(out, "obj"), exist_ok=True)
for spec in sys.argv[1:]:
    name, _, flags = spec.partition("=")
    lib = build.build(extra_flags=[f for f in flags.split(",") if f], lib=os.path.join(out, name + ".so"),
                      obj_dir=os.path.join(out, "obj", name))
    print(lib)
