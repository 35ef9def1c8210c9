#!/usr/bin/env python3
"""Variant builds of libdmnerf_hip.so for same-device A/B timing (tools/ab.sh): one library per -D flag set, objects
under tools/ab/obj/<name>/ (not shipped), libraries under tools/ab/<name>.so (git-ignored; they travel to the GPU box).

    python tools/ab_build.py base= scalar_sub=-DDMN_X3_SCALAR_SUB=1 ...

Only the sources whose include closure mentions one of the variant's macros are compiled for it; every other object is
the product build's (dm_nerf_amd/csrc/build/, brought up to date first)."""
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dm_nerf_amd import build  # noqa: E402


def closure_text(src: str, seen=None) -> str:
    seen = set() if seen is None else seen
    path = os.path.normpath(os.path.join(build.CSRC, src))
    if path in seen or not os.path.exists(path):
        return ""
    seen.add(path)
    text = open(path).read()
    for inc in re.findall(r'#include\s+"([^"]+)"', text):
        text += closure_text(os.path.join(os.path.dirname(src), inc), seen)
    return text


def main():
    out = os.path.join(ROOT, "tools", "ab")
    os.makedirs(os.path.join(out, "obj"), exist_ok=True)
    build.build()                                      # the product objects every variant starts from
    for spec in sys.argv[1:]:
        name, _, flags = spec.partition("=")
        flags = [f for f in flags.split(",") if f]
        macros = [re.sub(r"^-D", "", f).split("=")[0] for f in flags if f.startswith("-D")]
        obj_dir = os.path.join(out, "obj", name)
        os.makedirs(obj_dir, exist_ok=True)
        for src in build.SOURCES:
            if not any(m in closure_text(src) for m in macros):
                shutil.copy2(os.path.join(build.OBJ, src + ".o"), os.path.join(obj_dir, src + ".o"))
            elif os.path.exists(os.path.join(obj_dir, src + ".o")):
                os.remove(os.path.join(obj_dir, src + ".o"))      # flags may have changed under the same name
        lib = build.build(extra_flags=flags, lib=os.path.join(out, name + ".so"), obj_dir=obj_dir)
        print(lib, flush=True)


if __name__ == "__main__":
    main()
