"""The C-ABI library loads and exports every symbol the header declares (no GPU needed:
only host-side entry points are called here)."""
import ctypes
import os
import re

import pytest

import dm_nerf_amd
from dm_nerf_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "dmnerf_hip.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dmn_[a-z_0-9]+)\s*\(", text)))


def test_header_declares_the_expected_surface():
    names = declared_functions()
    for must in ("dmn_raygen", "dmn_encode", "dmn_field_fwd_rays", "dmn_field_fwd_embedded", "dmn_composite_fwd",
                 "dmn_sample_pdf", "dmn_resample_merge", "dmn_merge_depths", "dmn_jitter_depths", "dmn_field_pack"):
        assert must in names


def test_library_exports_every_declared_symbol_and_binding_matches():
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = declared_functions()
    for name in names:
        assert hasattr(lib, name), f"{name} declared in dmnerf_hip.h but not exported"
    assert sorted(_lib.SIGNATURES) == names, "ctypes table and header disagree"


def test_header_cites_reference_lines_for_every_compute_entry_point():
    text = open(HEADER).read()
    for anchor in ("networks/helpers.py:50-61", "networks/render.py:6-28", "networks/helpers.py:123-155",
                   "networks/dm_nerf.py:80-106", "networks/render.py:40-47", "networks/render.py:66-70",
                   "networks/dm_nerf.py:37-38", "manipulator.py:86-105"):
        assert anchor in text


def test_host_entry_points_without_gpu():
    lib = dm_nerf_amd.load_library()
    assert lib.dmn_abi_version() == _lib.ABI_VERSION
    assert lib.dmn_error_string(0) == b"ok"
    assert b"bad argument" in lib.dmn_error_string(-1)
    assert lib.dmn_field_param_floats(13) == 696338
    assert lib.dmn_field_param_floats(93) == 706658
    assert lib.dmn_field_blob_floats(13) == 3840 + 86 * 8192
    # argument validation happens on the host before any launch
    assert lib.dmn_field_pack_index(13, None) == -1
    assert lib.dmn_composite_fwd(None, None, 0, None, 4, 64, 18, 0, None, None, None, None, None) == -1
    assert lib.dmn_field_fwd_rays(None, None, None, 0, 4, 64, None, 13, None, None) == -1
    assert lib.dmn_field_fwd_rays(None, None, None, 0, 0, 64, None, 13, None, None) == 0      # empty input is a no-op
    assert lib.dmn_resample_merge(None, 0, None, None, 0, 0, 64, 128, None, None) == 0
    assert lib.dmn_sample_pdf(None, 0, None, 0, None, 0, 5, 1, 4, None, None) == -1           # n_bins < 2


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.HipLibraryError):
        _lib.load()


def test_cpu_tensors_are_refused_not_computed():
    import torch
    with pytest.raises(_lib.HipLibraryError):
        dm_nerf_amd.render_train(torch.zeros(2, 4, 18), torch.zeros(2, 4), torch.ones(2, 3))
    with pytest.raises(_lib.HipLibraryError):
        dm_nerf_amd.get_embedder(10, 0)[0].embed(torch.zeros(4, 3))


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "dm_nerf_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("# oracle", ""), f"{f} mentions the oracle"


def test_inline_asm_accumulator_reads_keep_their_distance_from_the_producing_mfma(tmp_path):
    """field_x3.h / field_x6.h read finished accumulator tiles with inline asm (v_accvgpr_read_b32) so that they stay in
    their AGPRs; the compiler inserts no MFMA -> VALU hazard wait states for inline asm, so the distance is guaranteed
    by construction (x3: one full k-step on the other tile; x6: 6 x (NT - 1) MFMAs) and checked here on the generated
    ISA of every kernel that does it -- the default (bf16x6, 16x16x32) field kernel, its 32x32x16 sibling, the bf16x6
    backward-data kernel and the fused kernel included.  The x6 kernels must also stay lean: no scratch (its traffic
    evicts the weight stream from the L2), no SGPRs parked in vector-register lanes (tools/check_acc_hazard.py)."""
    import shutil
    import subprocess
    import sys
    from concurrent.futures import ThreadPoolExecutor
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    from dm_nerf_amd.build import FLAGS
    flags = [f for f in FLAGS if f not in ("-shared", "-fPIC")]
    sources = {"field_bwd_x3.hip": [], "field_fwd_x3.hip": [], "field_fwd_x6.hip": ["--lean", "field_fwd_x6_kernel"],
               "field_fwd_x6w.hip": ["--lean", "field_fwd_x6w_kernel"], "field_bwd_x6.hip": ["--lean", "field_bwd_data_x6_kernel"],
               "field_fused_x3.hip": []}

    def check(src):
        out = str(tmp_path / (src + ".s"))
        subprocess.run([hipcc, *flags, "-S", "--cuda-device-only", "-o", out,
                        os.path.join(root, "dm_nerf_amd", "csrc", src)], check=True, capture_output=True)
        return subprocess.run([sys.executable, os.path.join(root, "tools", "check_acc_hazard.py"), out, *sources[src]],
                              capture_output=True, text=True)

    with ThreadPoolExecutor(max_workers=6) as pool:
        results = dict(zip(sources, pool.map(check, sources)))
    for src, res in results.items():
        assert res.returncode == 0, src + "\n" + res.stdout + res.stderr
        assert "accumulator reads" in res.stdout
    # every instantiation of the default kernel was seen (4 head widths x inference / saving), and of the fused x6 network
    assert results["field_fwd_x6.hip"].stdout.count("field_fwd_x6_kernel") == 8
    assert results["field_fwd_x6w.hip"].stdout.count("field_fwd_x6w_kernel") == 5      # the default inference kernel: 1, 2, 4, 6, 8 tiles
    assert results["field_bwd_x6.hip"].stdout.count("field_bwd_data_x6_kernel") == 4
    assert results["field_fused_x3.hip"].stdout.count("NetX6") == 4
