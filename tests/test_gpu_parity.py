"""GPU parity tests: the HIP path (through the C ABI) against the golden vectors produced by the
reference and against the CPU oracle on the same seeded inputs.

Tolerances (north-star: 1e-4 relative fp32, 'relative' = max-abs difference over the max magnitude
of the expected tensor):
  * ray generation, encodings:          <= 2 ulp-scale absolute (1e-6 / 2e-7)
  * field outputs (raw), rgb, ins maps:  1e-4 relative (observed ~1e-6)
  * depths after the hierarchical sampler (z_vals_fine, depth_fine): 1e-3 relative end to end.
    The reference's own fp32 path carries a ~3e-4 noise floor there (fp32-vs-fp64 run of the
    reference, BASELINE.md section 2: inverse-CDF sampling divides 1e-7 cdf differences by bin masses
    down to 1e-5).  The stage-wise tests feed the golden depths in and require 1e-4.
"""
import math

import numpy as np
import pytest
import torch

import cases
from conftest import rel_err
from oracle import dmnerf_cpu as oracle

import dm_nerf_amd as dm
from dm_nerf_amd import ops
from types import SimpleNamespace

pytestmark = pytest.mark.gpu
DEV = "cuda"


def G(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def H(t):
    return t.detach().cpu().numpy()


def nets(ins_num, seed_c, seed_f):
    args = dm.default_args(ins_num=ins_num)
    pos, view, coarse, fine, args = dm.create_nerf(args)
    coarse.load_state_dict(oracle.make_params(ins_num, seed_c))
    fine.load_state_dict(oracle.make_params(ins_num, seed_f))
    return pos, view, coarse.eval(), fine.eval(), args


def test_device_is_gfx950_and_library_is_loaded():
    import ctypes
    lib = dm.load_library()
    buf = ctypes.create_string_buffer(128)
    assert lib.dmn_device_name(buf, 128) == 0
    assert buf.value.decode().startswith("gfx950")


# ---- G1 -------------------------------------------------------------------------------------
def test_raygen_matches_reference_golden(golden):
    g = golden("g1_rays.npz")
    Hh, W, Ks, c2w = cases.camera_cases()
    for name, K in Ks.items():
        o, d = dm.get_rays_k(Hh, W, K, torch.from_numpy(c2w))
        assert o.shape == (Hh, W, 3) and d.shape == (Hh, W, 3)
        np.testing.assert_array_equal(H(o), g[f"{name}_o"])
        np.testing.assert_allclose(H(d), g[f"{name}_d"], rtol=0, atol=1e-6)
    o, d = dm.get_rays_k(Hh, W, Ks["dmsr"], c2w[:3, :4])
    np.testing.assert_allclose(H(d), g["dmsr34_d"], rtol=0, atol=1e-6)


def test_raygen_full_size_matches_oracle():
    Hh = W = 800
    K, c2w = oracle.dmsr_intrinsics(Hh, W), oracle.study_pose(3)
    o, d = dm.get_rays_k(Hh, W, K, c2w)
    oo, od = oracle.rays_from_camera(Hh, W, K, torch.from_numpy(c2w))
    np.testing.assert_array_equal(H(o), oo.numpy())
    np.testing.assert_allclose(H(d), od.numpy(), rtol=0, atol=1e-6)


# ---- G2 -------------------------------------------------------------------------------------
def test_encoder_matches_reference_golden(golden):
    g = golden("g2_encode.npz")
    x, v = cases.encoder_points()
    pos, _ = dm.get_embedder(10, 0)
    view, _ = dm.get_embedder(4, 0)
    ep, ev = H(pos.embed(G(x))), H(view.embed(G(v)))
    assert ep.shape == (256, 63) and ev.shape == (256, 27)
    # arguments reach 7.7e3 rad; both sides are ~1 ulp implementations
    np.testing.assert_allclose(ep, g["pos"], rtol=0, atol=2e-7)
    np.testing.assert_allclose(ev, g["dir"], rtol=0, atol=2e-7)
    np.testing.assert_array_equal(ep[:, :3], x)


# ---- G3 -------------------------------------------------------------------------------------
@pytest.mark.parametrize("ins_num,seed", [(13, 0), (93, 3), (7, 4)])
def test_field_forward_on_encodings_matches_reference_golden(golden, ins_num, seed):
    g = golden("g3_field.npz")
    x, v = cases.field_rows()
    emb = torch.cat([oracle.encode(torch.from_numpy(x), 10), oracle.encode(torch.from_numpy(v), 4)], -1)
    model = dm.DM_NeRF(8, 256, 63, 27, [4], ins_num).to(DEV)
    model.load_state_dict(oracle.make_params(ins_num, seed))
    with torch.no_grad():
        raw = H(model(emb.to(DEV)))
    assert raw.shape == (256, 5 + ins_num)
    assert rel_err(raw, g[f"raw_ins{ins_num}"]) < 1e-4
    assert rel_err(raw, g[f"raw_ins{ins_num}"]) < 1e-5       # what fp32 MFMA actually delivers


@pytest.mark.parametrize("n_rays,n_samples,ins_num", [(37, 64, 13), (21, 192, 13), (5, 100, 93), (3, 7, 40)])
def test_field_forward_on_rays_matches_oracle(n_rays, n_samples, ins_num):
    """Kernel-built points and encodings (ragged tile tails, S not a multiple of 32)."""
    rng = np.random.default_rng(100 + n_rays)
    o = rng.uniform(-1, 1, size=(n_rays, 3)).astype(np.float32)
    d = rng.standard_normal((n_rays, 3)).astype(np.float32)
    z = np.sort(rng.uniform(cases.NEAR, cases.FAR, size=(n_rays, n_samples)).astype(np.float32), -1)
    params = oracle.make_params(ins_num, 7)
    model = dm.DM_NeRF(8, 256, 63, 27, [4], ins_num).to(DEV)
    model.load_state_dict(params)
    raw = H(ops.field_rays(G(o), G(d), G(z), model.blob(), ins_num))
    with torch.no_grad():
        want = oracle.field_on_rays(params, torch.from_numpy(o), torch.from_numpy(d), torch.from_numpy(z)).numpy()
    assert raw.shape == want.shape == (n_rays, n_samples, 5 + ins_num)
    assert rel_err(raw, want) < 1e-4
    # shared lattice (stride-0 view) gives the same result as materialised rows
    lattice = torch.from_numpy(z[:1]).to(DEV).expand(n_rays, n_samples)
    raw_l = H(ops.field_rays(G(o), G(d), lattice, model.blob(), ins_num))
    raw_m = H(ops.field_rays(G(o), G(d), lattice.contiguous(), model.blob(), ins_num))
    np.testing.assert_array_equal(raw_l, raw_m)


def test_in_kernel_encodings_match_the_reference_encoder():
    """The field kernels build their own encodings (one fp64 sincos per coordinate, then the double-angle recurrence over
    the power-of-two bands).  The saving forward writes them into its record: read them back and hold them to the
    reference's Embedder (torch fp32 sin / cos of the fp32 argument) at the stand-alone encoder's tolerance, 2e-7 abs,
    on coordinates up to +-15 (arguments to 7.7e3 rad)."""
    rng = np.random.default_rng(77)
    n, s = 8, 32                                               # 8 tiles of 32 points
    o = rng.uniform(-3, 3, size=(n, 3)).astype(np.float32)
    d = rng.standard_normal((n, 3)).astype(np.float32)
    z = np.sort(rng.uniform(0.5, 8.0, size=(n, s)).astype(np.float32), -1)
    model = dm.DM_NeRF(8, 256, 63, 27, [4], 13).to(DEV)
    model.load_state_dict(oracle.make_params(13, 0))
    for precision in ("fp32", "bf16x6", "bf16x3"):
        _, save = ops.field_rays_save(G(o), G(d), G(z), model.blob(precision), 13, precision)
        rec = H(save).reshape(n, -1, 64)[:, :48]               # per tile: 32 position + 16 view registers x 64 lanes
        pts = (torch.from_numpy(o)[:, None, :] + torch.from_numpy(d)[:, None, :] * torch.from_numpy(z)[:, :, None])
        view = torch.from_numpy(d) / torch.norm(torch.from_numpy(d), dim=-1, keepdim=True)
        want_p = oracle.encode(pts.reshape(-1, 3), 10).numpy().reshape(n, s, 63)
        want_v = oracle.encode(view[:, None, :].expand(n, s, 3).reshape(-1, 3), 4).numpy().reshape(n, s, 27)
        assert 11.0 < np.abs(pts.numpy()).max() < 20.0
        worst = 0.0
        for h in range(2):
            for j in range(32):                                # field_layout.h pos_slot(j, h)
                col = (0 if h == 0 else 2) if j == 0 else ((1 if h == 0 else -1) if j == 1 else
                                                           3 + 6 * ((j - 2) // 3) + 3 * h + (j - 2) % 3)
                if col >= 0:
                    worst = max(worst, float(np.abs(rec[:, j, 32 * h:32 * h + 32] - want_p[:, :, col]).max()))
            for j in range(14):                                # dir_slot(j, h)
                col = (0 if h == 0 else 2) if j == 0 else ((1 if h == 0 else -1) if j == 1 else
                                                           3 + 6 * ((j - 2) // 3) + 3 * h + (j - 2) % 3)
                if col >= 0:
                    worst = max(worst, float(np.abs(rec[:, 32 + j, 32 * h:32 * h + 32] - want_v[:, :, col]).max()))
        assert worst <= 2e-7, (precision, worst)


def test_blob_follows_parameter_updates():
    model = dm.DM_NeRF(8, 256, 63, 27, [4], 13).to(DEV)
    model.load_state_dict(oracle.make_params(13, 0))
    b0 = model.blob()
    assert model.blob() is b0                                   # cached
    with torch.no_grad():
        model.density_linear.bias.add_(1.0)
    b1 = model.blob()
    assert b1 is not b0 and not torch.equal(b0, b1)
    # a write through .data bypasses autograd's version counter (by design): the cache cannot see it, invalidate_blobs() does
    model.density_linear.bias.data.add_(1.0)
    assert model.blob() is b1
    model.invalidate_blobs()
    b2 = model.blob()
    assert b2 is not b1 and not torch.equal(b1, b2)
    sd = {k: v.cpu() for k, v in model.state_dict().items()}
    assert list(sd) == [k for k, _ in oracle.param_specs(13)]   # checkpoint key contract


# ---- G4 -------------------------------------------------------------------------------------
@pytest.mark.parametrize("tag,shape", [("a", (32, 64, 18, 141)), ("b", (16, 192, 18, 142)), ("c", (8, 64, 98, 143)),
                                       ("d", (8, 100, 11, 144))])
def test_composite_matches_reference_golden(golden, tag, shape):
    g = golden("g4_composite.npz")
    raw, z, d = cases.composite_case(*shape)
    rgb, w, depth, ins = dm.render_train(G(raw), G(z), G(d))
    for name, val in (("rgb", rgb), ("w", w), ("depth", depth), ("ins", ins)):
        got, want = H(val), g[f"{tag}_{name}"]
        assert got.shape == want.shape, name
        assert rel_err(got, want) < 1e-4, (tag, name)
        assert rel_err(got, want) < 5e-6, (tag, name)
    # manipulator variant keeps the last channel (restated from text: parity unpinned by the reference)
    rgb2, w2, depth2, ins2 = dm.manipulator_render(G(raw), G(z), G(d))
    _, _, _, want_ins = oracle.composite(torch.from_numpy(raw), torch.from_numpy(z), torch.from_numpy(d),
                                         manipulator=True)
    assert ins2.shape[-1] == shape[2] - 4
    assert rel_err(H(ins2), want_ins.numpy()) < 5e-6
    np.testing.assert_array_equal(H(w2), H(w))


def test_composite_accepts_shared_lattice_and_empty_batch():
    raw, z, d = cases.composite_case(8, 64, 18, 141)
    z[:] = z[3]
    a = dm.render_train(G(raw), G(z), G(d))
    b = dm.render_train(G(raw), G(z[:1]).expand(8, 64), G(d))
    for x, y in zip(a, b):
        np.testing.assert_array_equal(H(x), H(y))
    e = dm.render_train(torch.zeros(0, 64, 18, device=DEV), torch.zeros(0, 64, device=DEV),
                        torch.zeros(0, 3, device=DEV))
    assert e[0].shape == (0, 3) and e[3].shape == (0, 13)


# ---- G5 -------------------------------------------------------------------------------------
def _sampler_conditioning(bins, w, u):
    """fp64 evaluation of the sampler's intermediate quantities per draw: bin mass `den`, bin width,
    and whether the draw sits within fp32 noise of one of the algorithm's discontinuities."""
    w = w.astype(np.float64) + 1e-5
    pdf = w / w.sum(-1, keepdims=True)
    cdf = np.concatenate([np.zeros_like(pdf[:, :1]), np.cumsum(pdf, -1)], -1)
    n, b = cdf.shape
    idx = np.stack([np.searchsorted(cdf[i], u[i].astype(np.float64), side="right") for i in range(n)])
    lo, hi = np.clip(idx - 1, 0, None), np.clip(idx, None, b - 1)
    rows = np.arange(n)[:, None]
    den = cdf[rows, hi] - cdf[rows, lo]
    width = np.abs(bins[rows, hi].astype(np.float64) - bins[rows, lo])
    knot_gap = np.abs(cdf[:, None, :] - u[:, :, None].astype(np.float64)).min(-1)
    fragile = (np.abs(den - 1e-5) < 1e-6) | (knot_gap < 3e-7) | (den < 1e-5)
    return np.maximum(den, 1e-5), width, fragile


def test_sampler_matches_reference_golden(golden):
    g = golden("g5_sampler.npz")
    bins, w, u = cases.sampler_case()
    det = H(dm.sample_pdf(G(bins), G(w), cases.N_IMPORTANCE, det=True))
    rnd = H(dm.sample_pdf(G(bins), G(w), cases.N_IMPORTANCE, det=False, u=G(u)))
    det40 = H(dm.sample_pdf(G(bins), G(w), 40, det=True))
    lin128 = torch.linspace(0, 1, 128).numpy()[None].repeat(len(bins), 0)
    lin40 = torch.linspace(0, 1, 40).numpy()[None].repeat(len(bins), 0)
    for got, key, draws in ((det, "det", lin128), (rnd, "rnd", u), (det40, "det40", lin40)):
        diff = np.abs(got - g[key])
        # The reference's sampler is itself discontinuous in fp32 (its fp64 run differs from its fp32 run
        # by a whole bin, 0.17, on 0.1% of these draws): `denom < 1e-5 -> 1` (helpers.py:150-151) flips on
        # 1e-7 noise when a bin's mass is ~1e-5, and t = (u - cdf_b) / denom magnifies cdf rounding by
        # 1/denom.  So: draws whose fp64 evaluation sits on such a threshold are excluded (must be rare),
        # every other draw must agree within the magnified rounding bound.
        den, width, fragile = _sampler_conditioning(bins, w, draws)
        assert fragile.mean() < 0.06, key       # det draws 0 and 1 sit on the end knots
        bound = 6e-7 / den * width + 4e-6
        bad = (diff > bound) & ~fragile
        assert not bad.any(), (key, float(diff[~fragile].max()), int(bad.sum()))
        assert np.median(diff) < 1e-6, key
    # the kernel's own linspace equals torch.linspace bit for bit
    lin = H(ops.sample_pdf(G(bins), G(w), cases.N_IMPORTANCE, None))
    np.testing.assert_array_equal(lin, det)
    lin40 = H(ops.sample_pdf(G(bins), G(w), 40, None))
    np.testing.assert_array_equal(lin40, det40)


def test_merge_is_sorted_and_is_a_permutation():
    rng = np.random.default_rng(5)
    a = np.sort(rng.uniform(4, 15, size=(33, 64)).astype(np.float32), -1)
    b = rng.uniform(4, 15, size=(33, 128)).astype(np.float32)
    b[0, :10] = a[0, :10]                                         # ties
    out = H(ops.merge_depths(G(a), G(b)))
    np.testing.assert_array_equal(out, np.sort(np.concatenate([a, b], -1), -1))
    # both runs ascending (the render path's case: lattice + monotone resampled depths): the merge by binary search,
    # with ties inside and across the runs, an empty run, and a NaN that sends one ray back to the rank sort
    bs = np.sort(b, -1)
    bs[1, 5:9] = bs[1, 5]
    bs[2, :3] = a[2, 7]
    out = H(ops.merge_depths(G(a), G(bs)))
    np.testing.assert_array_equal(out, np.sort(np.concatenate([a, bs], -1), -1))
    np.testing.assert_array_equal(H(ops.merge_depths(G(a), G(bs[:, :0]))), a)
    bn = bs.copy()
    bn[4, 17] = np.nan
    out = H(ops.merge_depths(G(a), G(bn)))
    np.testing.assert_array_equal(out[:4], np.sort(np.concatenate([a, bn], -1), -1)[:4])
    np.testing.assert_array_equal(out[4], np.sort(np.concatenate([a[4], bn[4]])))          # NaN last, as torch.sort


def test_jitter_matches_oracle():
    rng = np.random.default_rng(6)
    z = oracle.coarse_depths(19, cases.NEAR, cases.FAR, 64)
    t = torch.from_numpy(rng.uniform(0, 1, size=(19, 64)).astype(np.float32))
    got = H(ops.jitter_depths(dm.z_val_sample(19, cases.NEAR, cases.FAR, 64), t.to(DEV)))
    np.testing.assert_array_equal(got, oracle.jitter_depths(z, t).numpy())


# ---- G6 -------------------------------------------------------------------------------------
def _rays_off(got, want, tol=1e-4):
    """fraction of rays whose worst element misses `want` by more than tol * max|want|"""
    per_ray = np.abs(got - want).reshape(got.shape[0], -1).max(-1)
    return float((per_ray > tol * np.abs(want).max()).mean())


# End-to-end comparisons behind the sampler on SMALL ray samples (G6: 128 rays, G8: 480 pixels, spot checks of 48 / 64
# rays).  The yardstick is the reference's own behaviour, measured on the G13 sets (tests/golden/flipgate.py): run in fp64
# on the same rays the reference moves 0.98 - 1.07 % of them beyond 1e-4 (rgb / depth; none in the instance map, which is
# smooth in the depths).  Against the reference's fp32 output two such realisations differ on up to twice that many rays;
# the cap is 1.5x that (+ one ray: these samples hold one or two such rays).  The worst single ray is not capped here --
# it is the extreme of a heavy tail and differs from one fp32-grade arithmetic to the next -- the G13 tests prove for
# every ray beyond 1e-4 that a resampled depth moved and that the fine stage is within 1e-4 on the reference's depths.
REFERENCE_OWN_RATE = 0.0107


def _sampler_noise_ok(got, want, what=""):
    """Rays beyond 1e-4 of `want` are at most 3 x the reference's own fp32-vs-fp64 rate + 1; returns (count, worst)."""
    per_ray = np.abs(np.asarray(got, np.float64) - want).reshape(got.shape[0], -1).max(-1) / np.abs(want).max()
    off = int((per_ray > 1e-4).sum())
    assert off <= 3.0 * REFERENCE_OWN_RATE * got.shape[0] + 1, (what, off, got.shape[0], float(per_ray.max()))
    assert float(per_ray.max()) < 2e-2, (what, float(per_ray.max()))            # sanity: a moved sample, not a wrong ray
    return off, float(per_ray.max())


_Z_CAPS = dict(z_frac=0.003, z_max=2.8e-4)      # merged depths against the golden (G6): elements beyond 1e-4, worst


def _check_path(out, g, prefix, precision="fp32"):
    """End-to-end comparison of one dm_nerf call with the reference's output.

    Everything up to the sampler must hold 1e-4 outright (observed: 7e-7; held to 2e-6).  Behind the sampler the
    reference is not 1e-4-reproducible against ITSELF: run in fp64 on these very 128 rays it moves one ray's rgb_fine by
    2.3e-4 and depth_fine by 3.6e-4 of full scale, because `denom < 1e-5 -> 1` (helpers.py:150-151) makes
    a resampled depth jump by a bin width on 1e-7 noise.  So the fine outputs are held to the reference's own rate
    (_sampler_noise_ok), and test_fine_stage_on_golden_depths holds the fine stage itself (reference depths fed in) to
    1e-4 outright."""
    for k in ("rgb_coarse", "ins_coarse", "depth_coarse", "z_vals_coarse"):
        assert rel_err(H(out[k]), g[f"{prefix}_{k}"]) < 2e-6, k
    for k in ("rgb_fine", "depth_fine"):
        _sampler_noise_ok(H(out[k]), g[f"{prefix}_{k}"], (precision, prefix, k))
    assert rel_err(H(out["ins_fine"]), g[f"{prefix}_ins_fine"]) < 1e-4             # smooth in the depths: no ray leaves 1e-4
    got, want = H(out["z_vals_fine"]), g[f"{prefix}_z_vals_fine"]
    assert rel_err(got, want) < _Z_CAPS["z_max"]
    assert np.mean(np.abs(got - want) > 1e-4 * np.abs(want).max()) < _Z_CAPS["z_frac"]


def test_render_call_matches_reference_golden(golden):
    g = golden("g6_path.npz")
    n = 128
    o, d, t_rand, u = cases.path_rays(n)
    pos, view, coarse, fine, args = nets(13, 0, 1)
    rays = G(np.stack([o, d], 0))
    z = dm.z_val_sample(n, cases.NEAR, cases.FAR, cases.N_SAMPLES)
    np.testing.assert_array_equal(H(z[0]), g["z_lattice"])
    with torch.no_grad():
        det = dm.dm_nerf(rays, pos, view, coarse, fine, z, args)
        assert set(det) == {"rgb_fine", "ins_fine", "z_vals_fine", "raw_fine", "raw_coarse", "rgb_coarse",
                            "ins_coarse", "z_vals_coarse", "depth_fine", "depth_coarse"}
        assert det["raw_fine"].shape == (n, 192, 18) and det["raw_coarse"].shape == (n, 64, 18)
        _check_path(det, g, "det")
        assert rel_err(H(det["raw_coarse"][:8]), g["det_raw_coarse_head"]) < 1e-4
        # stochastic path with the reference's recorded draws
        args_r = SimpleNamespace(**{**vars(args), "perturb": 1.0, "is_train": True, "t_rand": G(t_rand), "u": G(u)})
        rnd = dm.dm_nerf(rays, pos, view, coarse, fine, z, args_r)
        _check_path(rnd, g, "rnd")
        # N_importance = 0: the fine network re-evaluates the coarse depths (BASELINE config 1)
        args_0 = SimpleNamespace(**{**vars(args), "N_importance": 0})
        ni0 = dm.dm_nerf(rays[:, :32], pos, view, coarse, fine, z[:32], args_0)
        assert ni0["z_vals_fine"].shape == (32, 64)
        for k in ("rgb_fine", "ins_fine", "depth_fine", "z_vals_fine"):      # no sampler draw involved
            assert rel_err(H(ni0[k]), g[f"ni0_{k}"]) < 1e-4, k
        # crop-sampler slice (render.py:88-90)
        args_n = SimpleNamespace(**{**vars(args), "is_train": True, "N_ins": 10})
        nins = dm.dm_nerf(rays[:, :32], pos, view, coarse, fine, z[:32], args_n)
        assert nins["ins_fine"].shape == (10, 13)
        assert _rays_off(H(nins["ins_fine"]), g["nins_ins_fine"]) <= 0.1


def test_fine_stage_on_golden_depths_is_within_1e4(golden):
    """Stage-wise: reference depths in, so the sampler's noise amplification is out of the picture."""
    g = golden("g6_path.npz")
    n = 128
    o, d, _, _ = cases.path_rays(n)
    _, _, coarse, fine, _ = nets(13, 0, 1)
    z_fine = G(g["det_z_vals_fine"])
    raw = ops.field_rays(G(o), G(d), z_fine, fine.blob(), 13)
    rgb, w, depth, ins = dm.render_train(raw, z_fine, G(d))
    assert rel_err(H(raw[:8]), g["det_raw_fine_head"]) < 1e-4
    assert rel_err(H(rgb), g["det_rgb_fine"]) < 1e-4
    assert rel_err(H(depth), g["det_depth_fine"]) < 1e-4
    assert rel_err(H(ins), g["det_ins_fine"]) < 1e-4


# ---- G8 -------------------------------------------------------------------------------------
@pytest.mark.parametrize("precision", [None, "fp32", "fp16x3", "bf16x3"])
def test_image_loop_with_ragged_tail_matches_reference_golden(golden, precision):
    """render_test on a 20x24 image in chunks of 100 (ragged tail of 80).  precision None = the image loop's default,
    which must be fp32-grade arithmetic: ops.DEFAULT_PRECISION ("bf16x6": every operand bit enters the products), held
    to the exact-fp32 kernel's caps -- never one of the narrower two-way splits."""
    g = golden("g8_image.npz")
    Hh, W = 20, 24
    pos, view, coarse, fine, args = nets(13, 0, 1)
    args.N_test = 100
    if precision is not None:
        args.precision = precision
    rgbs, inss, psnrs = dm.render_test(pos, view, coarse, fine, [cases.study_c2w()],
                                       [Hh, W, cases.study_intrinsics(Hh, W)], args, gt_imgs=[g["rgb"]])
    assert rgbs[0].shape == (Hh, W, 3) and inss[0].shape == (Hh, W, 13)
    assert ops.DEFAULT_PRECISION in ("bf16x6", "fp32")
    _sampler_noise_ok(H(rgbs[0]).reshape(-1, 3), g["rgb"].reshape(-1, 3), (precision, "image rgb"))
    assert rel_err(H(inss[0]).reshape(-1, 13), g["ins"].reshape(-1, 13)) < 1e-4
    assert psnrs[0] > 90.0                                     # PSNR of the HIP render against the reference's (94-99 dB)
    if precision is None:                                      # the default is ops.DEFAULT_PRECISION, bit for bit
        args.precision = ops.DEFAULT_PRECISION
        again, _, _ = dm.render_test(pos, view, coarse, fine, [cases.study_c2w()], [Hh, W, cases.study_intrinsics(Hh, W)],
                                     args)
        np.testing.assert_array_equal(H(again[0]), H(rgbs[0]))


# ---- BASELINE sizes: size-independent properties ------------------------------------------------
def test_full_size_launch_of_the_default_arithmetic_agrees_with_the_exact_kernel():
    """One launch of the bench's size (65536 rays x 192 samples = 12.6 M points through the fine network): the bf16x6
    kernel against the exact-fp32 MFMA kernel on every raw value -- both are fp32-grade, so they differ by accumulation
    rounding only --, deterministic run to run, and the density-only mode returns the same densities."""
    n, s = 65536, 192
    _, _, _, fine, _ = nets(13, 0, 1)
    rng = np.random.default_rng(123)
    o = G(rng.uniform(-1, 1, size=(n, 3)).astype(np.float32))
    d = G(rng.standard_normal((n, 3)).astype(np.float32))
    z = torch.sort(torch.rand(n, s, device=DEV) * (cases.FAR - cases.NEAR) + cases.NEAR, -1)[0]
    exact = ops.field_rays(o, d, z, fine.blob("fp32"), 13, "fp32")
    x6 = ops.field_rays(o, d, z, fine.blob("bf16x6"), 13, "bf16x6")
    scale = float(exact.abs().max())
    assert torch.isfinite(x6).all()
    worst = float((x6 - exact).abs().max()) / scale
    assert worst < 3e-6, worst                                   # observed ~1e-6 (fp32 accumulation order)
    assert torch.equal(x6, ops.field_rays(o, d, z, fine.blob("bf16x6"), 13, "bf16x6"))
    dens = ops.field_rays(o, d, z, fine.blob("bf16x6"), 13, "bf16x6", sigma_only=True)
    assert torch.equal(dens[..., 3], x6[..., 3])


def test_full_size_chunk_properties():
    Hh = W = 800
    pos, view, coarse, fine, args = nets(13, 0, 1)
    K, c2w = oracle.dmsr_intrinsics(Hh, W), oracle.study_pose(0)
    rays_o, rays_d = dm.get_rays_k(Hh, W, K, c2w)
    rays_o, rays_d = rays_o.reshape(-1, 3), rays_d.reshape(-1, 3)
    start, n = 317 * 800 + 123, 4096                            # a chunk from the middle of the image
    rays = torch.stack([rays_o[start:start + n], rays_d[start:start + n]], 0)
    z = dm.z_val_sample(n, args.near, args.far, args.N_samples)
    with torch.no_grad():
        full = dm.dm_nerf(rays, pos, view, coarse, fine, z, args)
        halves = [dm.dm_nerf(rays[:, i:i + 2048], pos, view, coarse, fine, z[:2048], args) for i in (0, 2048)]
    # partition invariance, bit for bit (what makes ray-sharding across GPUs exact)
    for k in ("rgb_fine", "ins_fine", "depth_fine", "z_vals_fine", "raw_fine"):
        np.testing.assert_array_equal(H(full[k]), np.concatenate([H(h[k]) for h in halves], 0))
    zf = H(full["z_vals_fine"])
    assert zf.shape == (n, 192) and np.all(np.diff(zf, axis=-1) >= 0)
    assert np.all(zf >= args.near) and np.all(zf <= args.far)
    lattice = H(z[0])
    assert all(np.isin(lattice, row).all() for row in zf[::257])     # coarse depths survive the merge
    _, w, _, _ = dm.render_train(full["raw_fine"], full["z_vals_fine"], rays[1])
    w = H(w)
    assert np.all(w >= 0) and np.all(w.sum(-1) <= 1 + 1e-5)
    assert np.all(np.isfinite(H(full["raw_fine"])))
    # spot-check 48 of the 4096 rays against the oracle
    pick = np.random.default_rng(9).choice(n, size=48, replace=False)
    with torch.no_grad():
        want = oracle.render_rays(rays[:, pick].cpu(), oracle.make_params(13, 0), oracle.make_params(13, 1),
                                  oracle.coarse_depths(48, args.near, args.far, 64))
    for k in ("rgb_fine", "depth_fine"):
        _sampler_noise_ok(H(full[k])[pick], want[k].numpy(), k)
    assert rel_err(H(full["ins_fine"])[pick], want["ins_fine"].numpy()) < 1e-4
    assert rel_err(H(full["rgb_coarse"])[pick], want["rgb_coarse"].numpy()) < 2e-6


def test_replica_shape_config():
    """BASELINE config 3 shape: 640x480 Replica intrinsics, ins_num = 93 (C = 98), near 0 / far 6.5."""
    Hh, W = 480, 640
    K = np.array([[W / 2.0, 0, (W - 1) / 2.0], [0, W / 2.0, (Hh - 1) / 2.0], [0, 0, 1]])
    pos, view, coarse, fine, args = nets(93, 3, 5)
    args.near, args.far = 0.0, 6.5
    c2w = oracle.study_pose(2)
    rays_o, rays_d = dm.get_rays_k(Hh, W, K, c2w)
    sel = np.random.default_rng(4).choice(Hh * W, size=96, replace=False)
    rays = torch.stack([rays_o.reshape(-1, 3)[sel], rays_d.reshape(-1, 3)[sel]], 0)
    z = dm.z_val_sample(96, args.near, args.far, 64)
    with torch.no_grad():
        got = dm.dm_nerf(rays, pos, view, coarse, fine, z, args)
        want = oracle.render_rays(rays.cpu(), oracle.make_params(93, 3), oracle.make_params(93, 5),
                                  oracle.coarse_depths(96, 0.0, 6.5, 64))
    assert got["raw_fine"].shape == (96, 192, 98) and got["ins_fine"].shape == (96, 93)
    for k in ("rgb_fine", "ins_fine"):
        assert _rays_off(H(got[k]), want[k].numpy()) <= 0.05, k
        assert rel_err(H(got[k]), want[k].numpy()) < 2e-3, k
    assert rel_err(H(got["ins_coarse"]), want["ins_coarse"].numpy()) < 1e-4
    assert rel_err(H(got["rgb_coarse"]), want["rgb_coarse"].numpy()) < 1e-4


# ---- end-to-end deviation behind the sampler, pinned to the REFERENCE's own behaviour (G13) ----------------------------
# The reference is not 1e-4-reproducible against itself behind sample_pdf (networks/helpers.py:139-153): run in fp64 on
# the same rays (same fp32 parameters) it moves ~1 % of them beyond 1e-4 -- `denom < 1e-5 -> 1` and the searchsorted
# bins make a resampled depth jump by a bin width on 1e-7 noise.  tests/golden/g13_flip.npz holds, for the 2048 random
# pixels of the 800x800 study camera and for smoke()'s 1024 rays: the rays, the reference's fp32 maps and merged depths,
# its fp64 maps, and its own per-ray fp32-vs-fp64 deviation.  Gates (every arithmetic):
#   (i)   the fraction of rays whose HIP fine outputs leave 1e-4 of the reference's fp64 result is at most 1.5x the
#         fraction of the reference's own fp32 result that does (same rays, same yardstick);
#   (ii)  attribution, 100 % of the rays: every ray beyond 1e-4 of the reference's fp32 maps has a merged depth that
#         differs from the reference's by more than 1e-4 -- a sampler flip -- and every ray, flipped or not, is within
#         1e-4 once the HIP fine stage is fed the reference's merged depths.
# Against the reference's fp32 maps themselves two independent flip sources add up (the reference's fp32 noise and
# ours): that fraction is reported and held to 2 x 1.5x the reference's own.
from flipgate import FLIP_KEYS as _FLIP_KEYS, flip_gate  # noqa: E402  (tests/golden/flipgate.py; smoke() runs the same gate)


def _flip_outputs(g, tag, precision):
    pos, view, coarse, fine, args = nets(13, 0, 1)
    args.precision = precision
    rays = G(g[f"{tag}_rays"])
    n = rays.shape[1]
    z = dm.z_val_sample(n, args.near, args.far, 64)
    with torch.no_grad():
        out = dm.dm_nerf(rays, pos, view, coarse, fine, z, args)
        z_ref = G(g[f"{tag}_z_vals_fine"])
        raw = ops.field_rays(rays[0], rays[1], z_ref, fine.blob(precision), 13, precision)
        rgb, _, depth, ins = dm.render_train(raw, z_ref, rays[1])
    got = {k: H(out[k]) for k in _FLIP_KEYS + ("z_vals_fine", "rgb_coarse", "depth_coarse")}
    return got, {"rgb_fine": H(rgb), "depth_fine": H(depth), "ins_fine": H(ins)}


@pytest.mark.parametrize("tag", ["t2048", "s1024"])
@pytest.mark.parametrize("precision", ["fp32", "fp16x3", "bf16x3", "bf16x6"])
def test_flip_rate_is_the_references_own_and_every_flip_is_a_sampler_flip(golden, precision, tag):
    g = golden("g13_flip.npz")
    got, staged = _flip_outputs(g, tag, precision)
    for k in ("rgb_coarse", "depth_coarse"):                          # in front of the sampler: 1e-4 outright
        assert rel_err(got[k], g[f"{tag}_{k}"]) < 1e-4, k
    report = flip_gate(got, g, tag, precision, staged)
    print(f"{precision} {tag}: {report}")


# ---- BASELINE config 5: ScanNet full-resolution render_path with the loader's crop mask, image-sharded ----------
def test_scannet_full_resolution_render_with_crop_mask():
    """968x1296 ScanNet colour intrinsics (4x4 matrix), near 0 / far 9.5, through render_test(..., crop_mask=...) as
    test_scannet.py:20-22 calls it: 19 chunks of 65536 rays + a ragged tail of 9344; the crop branch of
    tester.py:77-81 (select the mask's pixels, reshape to crop_height x crop_width) must be the centre rectangle of
    the full image bit for bit, and a sample of rays must match the CPU oracle."""
    Hh, W = cases.SCANNET_H, cases.SCANNET_W
    K = cases.scannet_intrinsics()
    ins_num = 22
    pos, view, coarse, fine, args = nets(ins_num, 3, 5)
    args.near, args.far = cases.SCANNET_NEAR, cases.SCANNET_FAR
    args.crop_width, args.crop_height = cases.SCANNET_CROP
    mask = cases.scannet_crop_mask(Hh, W, cases.SCANNET_CROP)
    assert int(mask.sum()) == 640 * 480 and (Hh * W) % 65536 == 9344
    pose = cases.scannet_pose(1)
    rgbs, inss, _ = dm.render_test(pos, view, coarse, fine, [pose], [Hh, W, K], args, crop_mask=mask, chunk=65536)
    assert rgbs[0].shape == (480, 640, 3) and inss[0].shape == (480, 640, ins_num)
    full_rgb, full_ins = dm.render_image(pos, view, coarse, fine, pose, [Hh, W, K], args, chunk=65536)
    assert full_rgb.shape == (Hh, W, 3)
    mh, mw = (Hh - 480) // 2, (W - 640) // 2
    np.testing.assert_array_equal(H(rgbs[0]), H(full_rgb[mh:Hh - mh, mw:W - mw]))
    np.testing.assert_array_equal(H(inss[0]), H(full_ins[mh:Hh - mh, mw:W - mw]))
    # flat mask (what the reference's indexing of [H*W, 3] needs) gives the same
    rgbs2, _, _ = dm.render_test(pos, view, coarse, fine, [pose], [Hh, W, K], args, crop_mask=mask.reshape(-1),
                                 chunk=4096 * 64)
    np.testing.assert_array_equal(H(rgbs2[0]), H(rgbs[0]))                       # chunking-invariant, too
    # a sample of rays (incl. the last pixels, which sit in the ragged tail) against the oracle
    pick = np.concatenate([np.random.default_rng(12).choice(Hh * W, size=60, replace=False),
                           np.arange(Hh * W - 4, Hh * W)])
    oo, od = oracle.rays_from_camera(Hh, W, K, torch.from_numpy(pose))
    rays = torch.stack([oo.reshape(-1, 3)[pick], od.reshape(-1, 3)[pick]], 0).float()
    with torch.no_grad():
        want = oracle.render_rays(rays, oracle.make_params(ins_num, 3), oracle.make_params(ins_num, 5),
                                  oracle.coarse_depths(len(pick), args.near, args.far, 64))
    got_rgb, got_ins = H(full_rgb).reshape(-1, 3)[pick], H(full_ins).reshape(-1, ins_num)[pick]
    print("scannet sample: rays off", _rays_off(got_rgb, want["rgb_fine"].numpy()), _rays_off(got_ins, want["ins_fine"].numpy()),
          "worst", rel_err(got_rgb, want["rgb_fine"].numpy()), rel_err(got_ins, want["ins_fine"].numpy()))
    _sampler_noise_ok(got_rgb, want["rgb_fine"].numpy(), "scannet rgb")
    assert rel_err(got_ins, want["ins_fine"].numpy()) < 1e-4


def test_pose_sharded_render_path_over_eight_poses():
    """Image-sharded rendering of config 5: the 8 poses dealt to 1, 2 and 8 ranks with parallel.pose_shard; every pose
    is rendered exactly once and a rank's images do not depend on which other poses it renders."""
    from dm_nerf_amd.parallel import pose_shard
    Hh, W = 30, 40
    K = cases.scannet_intrinsics().copy()
    K[:2, :3] *= W / cases.SCANNET_W
    pos, view, coarse, fine, args = nets(22, 3, 5)
    args.near, args.far, args.N_test = cases.SCANNET_NEAR, cases.SCANNET_FAR, 500
    poses = [cases.scannet_pose(i) for i in range(8)]
    whole, _, _ = dm.render_test(pos, view, coarse, fine, poses, [Hh, W, K], args)
    for world in (2, 8):
        seen = []
        for rank in range(world):
            mine = pose_shard(len(poses), rank, world)
            seen += mine
            rgbs, _, _ = dm.render_test(pos, view, coarse, fine, [poses[i] for i in mine], [Hh, W, K], args)
            for i, img in zip(mine, rgbs):
                np.testing.assert_array_equal(H(img), H(whole[i]))
        assert sorted(seen) == list(range(8))


# ---- split-precision field kernels (3 x 16-bit MFMA per product) -------------------------------------
@pytest.mark.parametrize("precision,tol", [("fp16x3", 2e-5), ("bf16x3", 1e-4), ("bf16x6", 4e-6)])
@pytest.mark.parametrize("ins_num,seed", [(13, 0), (93, 3), (7, 4)])
def test_split_precision_field_matches_reference_golden(golden, precision, tol, ins_num, seed):
    g = golden("g3_field.npz")
    x, v = cases.field_rows()
    emb = torch.cat([oracle.encode(torch.from_numpy(x), 10), oracle.encode(torch.from_numpy(v), 4)], -1)
    model = dm.DM_NeRF(8, 256, 63, 27, [4], ins_num).to(DEV)
    model.load_state_dict(oracle.make_params(ins_num, seed))
    model.precision = precision
    ops.take_status(DEV)
    with torch.no_grad():
        raw = H(model(emb.to(DEV)))
    err = rel_err(raw, g[f"raw_ins{ins_num}"])
    print(f"{precision} ins_num={ins_num}: rel err {err:.2e}")
    assert err < tol
    assert ops.take_status(DEV) == 0


@pytest.mark.parametrize("ins_num,seed", [(13, 0), (93, 3)])
def test_full_precision_bf16_kernel_is_as_close_to_fp64_as_the_fp32_paths(golden, ins_num, seed):
    """What makes "bf16x6" the reference's arithmetic and not a reduced one: against an fp64 evaluation of the same fp32
    parameters on the G3 rows, its raw output is no further away than (i) the reference's own fp32 result (the golden)
    and (ii) the exact-fp32 MFMA kernel -- all three sit on the rounding noise of fp32 accumulation -- while the two-way
    split of bf16x3 (16 operand bits) is an order of magnitude further out."""
    g = golden("g3_field.npz")
    x, v = cases.field_rows()
    emb = torch.cat([oracle.encode(torch.from_numpy(x), 10), oracle.encode(torch.from_numpy(v), 4)], -1)
    params = oracle.make_params(ins_num, seed)
    with torch.no_grad():
        want64 = oracle.field({k: p.double() for k, p in params.items()}, emb.double()).numpy()
    model = dm.DM_NeRF(8, 256, 63, 27, [4], ins_num).to(DEV)
    model.load_state_dict(params)
    err = {"reference fp32 (golden)": rel_err(g[f"raw_ins{ins_num}"], want64)}
    for precision in ("fp32", "bf16x6", "bf16x3"):
        model.precision = precision
        with torch.no_grad():
            err[precision] = rel_err(H(model(emb.to(DEV))), want64)
    print(f"ins_num={ins_num}: max |raw - fp64| / max |fp64|: " + ", ".join(f"{k} {e:.2e}" for k, e in err.items()))
    floor = max(err["reference fp32 (golden)"], err["fp32"])
    assert err["bf16x6"] <= 1.5 * floor, err
    assert err["bf16x6"] < 2e-6, err
    assert err["bf16x3"] > 3.0 * err["bf16x6"], err          # the two-way split is the one that is measurably narrower


@pytest.mark.parametrize("precision", ["fp16x3", "bf16x3", "bf16x6"])
def test_split_precision_on_rays_and_whole_call(golden, precision):
    g = golden("g6_path.npz")
    n = 128
    o, d, _, _ = cases.path_rays(n)
    pos, view, coarse, fine, args = nets(13, 0, 1)
    args.precision = precision
    rays = G(np.stack([o, d], 0))
    z = dm.z_val_sample(n, cases.NEAR, cases.FAR, cases.N_SAMPLES)
    with torch.no_grad():
        out = dm.dm_nerf(rays, pos, view, coarse, fine, z, args)
    _check_path(out, g, "det", precision)
    assert rel_err(H(out["raw_coarse"][:8]), g["det_raw_coarse_head"]) < 1e-4
    # fine stage on the reference's depths: 1e-4 outright
    z_fine = G(g["det_z_vals_fine"])
    raw = ops.field_rays(G(o), G(d), z_fine, fine.blob(precision), 13, precision)
    rgb, w, depth, ins = dm.render_train(raw, z_fine, G(d))
    for got, key in ((raw[:8], "det_raw_fine_head"), (rgb, "det_rgb_fine"), (depth, "det_depth_fine"),
                     (ins, "det_ins_fine")):
        assert rel_err(H(got), g[key]) < 1e-4, key
    # ragged tile tail and odd sample count
    params = oracle.make_params(13, 1)
    zz = G(g["det_z_vals_fine"][:5, :101])
    raw = H(ops.field_rays(G(o[:5]), G(d[:5]), zz, fine.blob(precision), 13, precision))
    with torch.no_grad():
        want = oracle.field_on_rays(params, torch.from_numpy(o[:5]), torch.from_numpy(d[:5]), zz.cpu()).numpy()
    assert rel_err(raw, want) < 1e-4


@pytest.mark.parametrize("precision", ["fp32", "fp16x3", "bf16x3", "bf16x6"])
@pytest.mark.parametrize("ins_num", [0, 31, 32, 70, 100, 127])
def test_every_instance_head_width_and_tiny_batches(precision, ins_num):
    """All four ins_linear tile-count instantiations (1..4 tiles of 32 channels), single-ray and single-sample
    launches, in every precision."""
    rng = np.random.default_rng(ins_num)
    params = oracle.make_params(ins_num, 11)
    model = dm.DM_NeRF(8, 256, 63, 27, [4], ins_num).to(DEV)
    model.load_state_dict(params)
    for n_rays, n_samples in ((1, 1), (2, 33), (1, 130)):
        o = rng.uniform(-1, 1, size=(n_rays, 3)).astype(np.float32)
        d = rng.standard_normal((n_rays, 3)).astype(np.float32)
        z = np.sort(rng.uniform(cases.NEAR, cases.FAR, size=(n_rays, n_samples)).astype(np.float32), -1)
        raw = H(ops.field_rays(G(o), G(d), G(z), model.blob(precision), ins_num, precision))
        with torch.no_grad():
            want = oracle.field_on_rays(params, torch.from_numpy(o), torch.from_numpy(d), torch.from_numpy(z)).numpy()
        assert raw.shape == (n_rays, n_samples, 5 + ins_num)
        assert rel_err(raw, want) < 1e-4, (n_rays, n_samples)
    assert ops.take_status(DEV) == 0


def test_fp16_range_flag_and_exact_fallback():
    """Activations beyond the fp16 range must be flagged and the image re-rendered exactly."""
    pos, view, coarse, fine, args = nets(13, 0, 1)
    big = oracle.make_params(13, 0)
    big["mlps.0.weight"] = big["mlps.0.weight"] * 3e4          # trunk activations ~1e5 > 65504
    coarse.load_state_dict(big)
    Hh, W = 8, 8
    hwk = [Hh, W, cases.study_intrinsics(Hh, W)]
    args.precision = "fp32"
    exact_rgb, exact_ins = dm.render_image(pos, view, coarse, fine, cases.study_c2w(), hwk, args)
    args.precision = "fp16x3"
    with pytest.warns(UserWarning, match="fp16x3 range exceeded"):
        rgb, ins = dm.render_image(pos, view, coarse, fine, cases.study_c2w(), hwk, args)
    np.testing.assert_array_equal(H(rgb), H(exact_rgb))
    np.testing.assert_array_equal(H(ins), H(exact_ins))
    args.precision = "bf16x3"                                   # bf16 has fp32's range: no flag, close result
    rgb_b, _ = dm.render_image(pos, view, coarse, fine, cases.study_c2w(), hwk, args)
    assert np.isfinite(H(rgb_b)).all()
    # bf16x6 splits every fp32 value exactly whatever its magnitude: the coarse field on activations of ~1e5 stays at
    # the exact kernel's distance from the oracle
    o, d, _, _ = cases.path_rays(24, seed=3)
    z = dm.z_val_sample(24, cases.NEAR, cases.FAR, 64)
    with torch.no_grad():
        want = oracle.field_on_rays(big, torch.from_numpy(o), torch.from_numpy(d), z.cpu().contiguous()).numpy()
    err = {p: rel_err(H(ops.field_rays(G(o), G(d), z, coarse.blob(p), 13, p)), want) for p in ("fp32", "bf16x6")}
    assert err["bf16x6"] < 2.0 * err["fp32"] + 1e-7 and err["bf16x6"] < 1e-5, err


# ---- training: gradients through the HIP path (row T / K11) --------------------------------------------
def test_composite_backward_matches_oracle_autograd():
    for (n, s, c, seed) in ((24, 64, 18, 141), (9, 192, 18, 142), (6, 100, 11, 144), (6, 64, 98, 143)):
        raw, z, d = cases.composite_case(n, s, c, seed)
        rng = np.random.default_rng(seed + 1)
        probes = [rng.standard_normal(sh).astype(np.float32) for sh in ((n, 3), (n, s), (n,), (n, c - 5))]
        raw_c = torch.from_numpy(raw).requires_grad_(True)
        outs = oracle.composite(raw_c, torch.from_numpy(z), torch.from_numpy(d))
        sum((o * torch.from_numpy(p)).sum() for o, p in zip(outs, probes)).backward()
        raw_g = G(raw).requires_grad_(True)
        outs_g = dm.render_train(raw_g, G(z), G(d))
        sum((o * G(p)).sum() for o, p in zip(outs_g, probes)).backward()
        err = rel_err(H(raw_g.grad), raw_c.grad.numpy())
        assert err < 1e-4, (n, s, c, err)
        # each gradient source on its own (NULL pointers for the others)
        for k in range(4):
            raw_g = G(raw).requires_grad_(True)
            (dm.render_train(raw_g, G(z), G(d))[k] * G(probes[k])).sum().backward()
            raw_c = torch.from_numpy(raw).requires_grad_(True)
            (oracle.composite(raw_c, torch.from_numpy(z), torch.from_numpy(d))[k] * torch.from_numpy(probes[k])).sum().backward()
            scale = max(float(raw_c.grad.abs().max()), 1e-30)
            assert float(np.abs(H(raw_g.grad) - raw_c.grad.numpy()).max()) < 1e-4 * scale, (k, n, s, c)


@pytest.mark.parametrize("precision", ["fp32", "bf16x6", "fp16x3"])
def test_training_gradients_match_reference_golden(golden, precision):
    """One backward pass through dm_nerf (stochastic path, recorded draws) against the reference's gradients."""
    g = golden("g7_gradients.npz")
    n = 48
    o, d, t_rand, u = cases.path_rays(n, seed=18)
    pos, view, coarse, fine, args = nets(13, 0, 1)
    coarse.train(); fine.train()
    args = SimpleNamespace(**{**vars(args), "perturb": 1.0, "is_train": True, "t_rand": G(t_rand), "u": G(u),
                              "precision": precision})
    z = dm.z_val_sample(n, cases.NEAR, cases.FAR, cases.N_SAMPLES)
    res = dm.dm_nerf(G(np.stack([o, d], 0)), pos, view, coarse, fine, z, args)
    assert res["rgb_fine"].requires_grad and res["raw_coarse"].requires_grad and not res["z_vals_fine"].requires_grad
    probe = cases.probe_weights({k: tuple(res[k].shape) for k in cases.PROBE_KEYS})
    loss = sum((res[k] * G(probe[k])).sum() for k in cases.PROBE_KEYS)
    loss.backward()
    assert abs(float(loss) - float(g["loss"])) < 2e-3 * abs(float(g["loss"]))   # sampler noise floor, see _check_path
    worst = 0.0
    for tag, net in (("c", coarse), ("f", fine)):
        norms = np.array([np.linalg.norm(H(p.grad).astype(np.float64)) for p in net.parameters()])
        np.testing.assert_allclose(norms, g[f"{tag}.norms"], rtol=5e-3)
        for name, p in net.named_parameters():
            got = H(p.grad)[cases.grad_digest_slices(name, p.shape)]
            err = rel_err(got, g[f"{tag}.{name}"])
            l2 = float(np.linalg.norm((got - g[f"{tag}.{name}"]).astype(np.float64))
                       / max(np.linalg.norm(g[f"{tag}.{name}"].astype(np.float64)), 1e-30))
            worst = max(worst, err)
            # Observed on MI355X (max-abs / max-magnitude, and relative L2), caps at ~2x:
            #   exact forward: coarse network 2.1e-5 / 1.6e-5 -- it does not see the sampler, held to 5e-5.  The fine
            #     network is evaluated at the resampled depths, where the reference disagrees with ITSELF by 3.5e-2
            #     (its fp64 run against these fp32 goldens; cumprod's backward divides by 1 - alpha + 1e-10):
            #     9.8e-3 / 6.9e-3 observed; test_fine_stage_gradients_on_golden_depths holds that stage alone to 5e-3.
            #   split-precision forward (bf16x3, 7e-6 on the activations): a handful of ReLUs of the 3072 points sit on
            #     the other side of zero and the compositing gradient concentrates on few samples per ray, so the
            #     gradients that pass through a ReLU mask of this network move by up to 2.5e-2 in single entries
            #     (6.9e-3 in L2; fine network 2.9e-2 / 2.0e-2).  The parameters whose gradient passes through no mask
            #     of the perturbed forward (the detached instance branch, the density and colour heads) keep 1e-5 and
            #     are held to 1e-4; test_field_backward_kernels_match_library_gemm_backward holds the masked ones to
            #     1e-4 away from units at zero.
            mask_free = name.startswith(("ins_", "density_linear", "rgb_linear"))
            if precision in ("fp32", "bf16x6"):         # fp32-grade forwards (exact MFMA / exact three-way bf16 splits)
                cap_max, cap_l2 = (5e-5, 5e-5) if tag == "c" else (2e-2, 1.4e-2)
            elif tag == "c":
                cap_max, cap_l2 = (1e-4, 1e-4) if mask_free else (5e-2, 1.4e-2)
            else:
                cap_max, cap_l2 = 6e-2, 4e-2
            assert err < cap_max and l2 < cap_l2, (tag, name, err, l2)
    print(f"{precision}: worst gradient rel err {worst:.2e}")
    # an optimizer step through the standard torch API works on the module's parameters (train_dmsr.py:127-128)
    opt = torch.optim.Adam(list(coarse.parameters()) + list(fine.parameters()), lr=5e-4, betas=(0.9, 0.999))
    before = coarse.blob().clone()
    opt.step()
    assert not torch.equal(before, coarse.blob())               # the packed weights follow the update


def test_training_steps_reduce_loss_and_checkpoint_roundtrip(tmp_path):
    """BASELINE config 4 shape in small: stochastic path, Adam(5e-4), lr decay after the step, reference
    checkpoint keys (train_dmsr.py:82-87,127-128)."""
    from dm_nerf_amd import trainer
    pos, view, coarse, fine, args = nets(13, 0, 1)
    coarse.train(); fine.train()
    args.perturb, args.is_train = 1.0, True
    n = 256
    o, d, _, _ = cases.path_rays(n, seed=21)
    rays = G(np.stack([o, d], 0))
    target = torch.full((n, 3), 0.25, device=DEV)
    z = dm.z_val_sample(n, cases.NEAR, cases.FAR, cases.N_SAMPLES)
    opt = trainer.make_optimizer(coarse, fine, lrate=5e-4)
    torch.manual_seed(3)
    hist = [trainer.train_step(i, rays, target, (pos, view, coarse, fine, args), z, opt) for i in range(6)]
    assert all(np.isfinite(h["total"]) for h in hist)
    assert hist[-1]["total"] < hist[0]["total"]
    assert abs(opt.param_groups[0]["lr"] - 5e-4 * 0.1 ** (5 / 500000)) < 1e-12
    path = str(tmp_path / "000006.tar")
    torch.save(trainer.checkpoint_dict(6, coarse, fine, opt), path)
    ckpt = torch.load(path, map_location="cpu", weights_only=True)
    assert set(ckpt) == {"iteration", "network_coarse_state_dict", "network_fine_state_dict", "optimizer_state_dict"}
    _, _, coarse2, fine2, _ = nets(13, 5, 6)
    assert trainer.load_checkpoint(ckpt, coarse2, fine2) == 6
    with torch.no_grad():
        args.perturb, args.is_train = 0.0, False
        a = dm.dm_nerf(rays[:, :32], pos, view, coarse, fine, z[:32], args)["rgb_fine"]
        b = dm.dm_nerf(rays[:, :32], pos, view, coarse2, fine2, z[:32], args)["rgb_fine"]
    np.testing.assert_array_equal(H(a), H(b))


def test_train_step_with_all_loss_heads_is_the_same_step_in_every_calling_form():
    """The three ways of running a step with the instance and emptiness heads -- two ins_criterion calls (the reference's
    form, train_dmsr.py:38-45), ``trainer.instance_loss(...).many`` (both maps behind one wait for the device) and
    ``report=False`` (no read-back) -- give the same loss terms and leave bit-identical weights behind."""
    from dm_nerf_amd import trainer
    n = 192
    o, d, _, _ = cases.path_rays(n, seed=23)
    rays = G(np.stack([o, d], 0))
    target = torch.full((n, 3), 0.4, device=DEV)
    labels = torch.from_numpy(np.random.default_rng(5).integers(0, 13, size=n)).to(DEV)
    z = dm.z_val_sample(n, cases.NEAR, cases.FAR, cases.N_SAMPLES)
    results = []
    for form in ("two_calls", "many", "no_report"):
        pos, view, coarse, fine, args = nets(13, 0, 1)
        coarse.train(); fine.train()
        args.perturb, args.is_train, args.tolerance, args.deta_w = 1.0, True, 0.05, 0.05
        opt = trainer.make_optimizer(coarse, fine, lrate=5e-4)
        ins = (lambda pred: dm.ins_criterion(pred, labels, 13)[0]) if form == "two_calls" else trainer.instance_loss(labels, 13)
        pen = lambda raw, zz, depth: dm.ins_penalizer(raw, zz, depth, rays[1], args)
        torch.manual_seed(11)
        rep = None
        for i in range(2):
            rep = trainer.train_step(i, rays, target, (pos, view, coarse, fine, args), z, opt, ins_loss=ins, penalizer=pen,
                                     report=form != "no_report")
        rep = {k: float(v) for k, v in rep.items() if k != "psnr_fine"}
        results.append((rep, torch.cat([p.detach().reshape(-1) for p in list(coarse.parameters()) + list(fine.parameters())])))
    for rep, w in results[1:]:
        assert rep == results[0][0]
        assert torch.equal(w, results[0][1])


def test_fine_stage_gradients_on_golden_depths(golden):
    """Fine field + compositing backward at the reference's own resampled depths (sampler out of the loop)."""
    from dm_nerf_amd.autograd import FieldFn
    g = golden("g7_gradients.npz")
    n = 48
    o, d, _, _ = cases.path_rays(n, seed=18)
    _, _, _, fine, _ = nets(13, 0, 1)
    z_fine = G(g["z_vals_fine"])
    raw = FieldFn.apply(G(o), G(d), z_fine, fine, "fp32", *fine.parameters())
    rgb, _, depth, ins = dm.render_train(raw, z_fine, G(d))
    shapes = {"rgb_fine": (n, 3), "ins_fine": (n, 13), "depth_fine": (n,), "rgb_coarse": (n, 3), "ins_coarse": (n, 13),
              "depth_coarse": (n,), "raw_fine": (n, 192, 18), "raw_coarse": (n, 64, 18)}
    probe = cases.probe_weights(shapes)
    ((rgb * G(probe["rgb_fine"])).sum() + (ins * G(probe["ins_fine"])).sum() + (depth * G(probe["depth_fine"])).sum()
     + (raw * G(probe["raw_fine"])).sum()).backward()
    for name, p in fine.named_parameters():
        got = H(p.grad)[cases.grad_digest_slices(name, p.shape)]
        # cumprod's backward divides by 1 - alpha + 1e-10, which magnifies the 1e-6 forward differences of
        # saturated samples; observed 2.3e-3 worst (mlps.6.weight), 5e-3 allowed
        assert rel_err(got, g[f"f.{name}"]) < 5e-3, name


def _min_abs_preactivation(net, o, d, z):
    """Smallest |pre-activation| over all ReLU units per sample point (torch ops on the device, dm_nerf.py:80-106)."""
    import torch.nn.functional as F
    from library_backward import _encode
    p = dict(net.named_parameters())
    with torch.no_grad():
        view = d / torch.norm(d, dim=-1, keepdim=True)
        pts = o[:, None, :] + d[:, None, :] * z[:, :, None]
        pos = _encode(pts.reshape(-1, 3), 10)
        vdir = _encode(view[:, None, :].expand(pts.shape).reshape(-1, 3), 4)
        h, low = pos, None
        for i in range(8):
            a = F.linear(h, p[f"mlps.{i}.weight"], p[f"mlps.{i}.bias"])
            m = a.abs().min(-1).values
            low = m if low is None else torch.minimum(low, m)
            h = F.relu(a)
            if i == 4:
                h = torch.cat([h, pos], -1)
        feat = F.linear(h, p["rgb_feature_linear.weight"], p["rgb_feature_linear.bias"])
        a = F.linear(torch.cat([feat, vdir], -1), p["rgb_feature_linears.0.weight"], p["rgb_feature_linears.0.bias"])
        low = torch.minimum(low, a.abs().min(-1).values)
        code = F.linear(h, p["ins_feature_linear.weight"], p["ins_feature_linear.bias"])
        a = F.linear(code, p["ins_feature_linears.0.weight"], p["ins_feature_linears.0.bias"])
        low = torch.minimum(low, a.abs().min(-1).values)
    return low.reshape(z.shape)


@pytest.mark.parametrize("fwd,data,wgrad", [("fp32", "fp32", "fp32"), ("fp32", "fp32", "bf16x3"),
                                            ("fp32", "bf16x3", "bf16x3"), ("bf16x3", "bf16x3", "bf16x3"),
                                            ("bf16x6", "fp32", "fp32"), ("bf16x6", "bf16x3", "bf16x3"),
                                            ("bf16x6", "bf16x6", "bf16x6"), ("fp32", "bf16x6", "bf16x6"),
                                            ("bf16x6", "bf16x6", "fp32"), ("bf16x6", "fp32", "bf16x6")])
@pytest.mark.parametrize("ins_num,n,s", [(13, 37, 64), (93, 11, 100), (40, 5, 192), (13, 700, 64), (127, 3, 64)])
def test_field_backward_kernels_match_library_gemm_backward(ins_num, n, s, fwd, data, wgrad, monkeypatch):
    """HIP backward (save record -> backward-data kernel -> split-K weight-gradient kernel) against autograd
    through library GEMMs on the device, every parameter of the network, ragged point counts, 1..4 logit tiles,
    every combination of forward / backward-data / weight-gradient arithmetic that ships."""
    from dm_nerf_amd import autograd
    from dm_nerf_amd.autograd import FieldFn
    from library_backward import FieldFnLibrary
    monkeypatch.setattr(autograd, "BACKWARD_OVERRIDE", (data, wgrad))
    _, _, net, _, _ = nets(ins_num, 3, 4)
    rng = np.random.default_rng(900 + ins_num + n)
    o, d, _, _ = cases.path_rays(n, seed=31 + n)
    z = np.sort(rng.uniform(cases.NEAR, cases.FAR, (n, s)).astype(np.float32), -1)
    g_raw = rng.standard_normal((n, s, 5 + ins_num)).astype(np.float32)
    # A ReLU whose pre-activation is within rounding of zero may open in one forward and close in the other; that
    # is a different function, not an error.  Points with such a unit get no gradient in this comparison.
    # (the bf16x3 forward differs from torch's by ~1e-5 of the activation scale, the exact one by ~1e-6; the thresholds
    # are 3x / 10x that)
    low = H(_min_abs_preactivation(net, G(o), G(d), G(z)))
    # observed: 12..18 % of the points have a unit within 1e-5 of zero, 42 % within 3e-5 (2816 ReLU units per point)
    fp32_grade = fwd in ("fp32", "bf16x6")
    fragile = low < (1e-5 if fp32_grade else 3e-5)
    assert fragile.mean() < (0.25 if fp32_grade else 0.5)
    g_raw[fragile] = 0.0
    grads = {}
    for tag, fn in (("hip", FieldFn), ("lib", FieldFnLibrary)):
        net.zero_grad()
        raw = fn.apply(G(o), G(d), G(z), net, fwd, *net.parameters())
        (raw * G(g_raw)).sum().backward()
        grads[tag] = {k: H(p.grad).copy() for k, p in net.named_parameters()}
        if tag == "hip":
            raw_hip = H(raw)
        else:
            np.testing.assert_array_equal(raw_hip, H(raw))          # the saving forward is the same kernel
    # fp32-grade backward kernels (exact fp32 MFMA, or six bf16 products on exact three-way splits) are held to the
    # rounding of an fp32 reduction over the points; the two-way split kernels (16 operand bits) to the north-star's 1e-4
    # (observed: 1.1e-6 .. 2.1e-6 up to 2 368 points in every fp32-grade combination, 4.0e-6 (all exact) .. 6.1e-6 (all
    # bf16x6) at 44 800 points, where the order of the fp32 reduction over the points is what differs from the library's;
    # two-way splits 7e-6 .. 1.4e-5)
    bound = (5e-6 if n * s < 10000 else 8e-6) if data in ("fp32", "bf16x6") and wgrad in ("fp32", "bf16x6") else 1e-4
    worst = 0.0
    for k in grads["lib"]:
        err = rel_err(grads["hip"][k], grads["lib"][k])
        worst = max(worst, err)
        assert err < bound, (k, err)
    print(f"ins_num={ins_num} M={n * s} forward={fwd} data={data} wgrad={wgrad}: worst rel err vs library backward {worst:.2e}")


def test_field_backward_is_linear_and_additive_over_rays_at_training_size():
    """BASELINE config 4 size (4096 rays x 192 samples through the fine network): the parameter gradient is linear in
    dL/d raw and additive over a partition of the rays -- size-independent properties of the split-K reduction."""
    from dm_nerf_amd.autograd import FieldFn
    _, _, net, _, _ = nets(13, 3, 4)
    n, s = 4096, 192
    rng = np.random.default_rng(77)
    o, d, _, _ = cases.path_rays(n, seed=77)
    z = G(np.sort(rng.uniform(cases.NEAR, cases.FAR, (n, s)).astype(np.float32), -1))
    g_raw = G(rng.standard_normal((n, s, 18)).astype(np.float32))

    def grads(rows, scale=1.0):
        net.zero_grad()
        raw = FieldFn.apply(G(o)[rows], G(d)[rows], z[rows], net, "fp32", *net.parameters())
        (raw * (scale * g_raw[rows])).sum().backward()
        return torch.cat([p.grad.reshape(-1) for p in net.parameters()]).double()

    whole = grads(slice(0, n))
    assert torch.isfinite(whole).all() and float(whole.abs().max()) > 0
    scale = float(whole.abs().max())
    assert float((grads(slice(0, n), -2.0) + 2.0 * whole).abs().max()) < 2e-5 * scale          # linear (bf16x3 rounding)
    parts = grads(slice(0, 1000)) + grads(slice(1000, 3333)) + grads(slice(3333, n))
    assert float((parts - whole).abs().max()) < 2e-5 * scale                                   # additive
    assert torch.equal(grads(slice(0, n)), whole)                                              # deterministic


def test_field_backward_respects_detach_and_frozen_parameters():
    """dm_nerf.py:95: the instance branch reads h.detach() -- a loss on the instance logits alone gives the trunk
    no gradient; parameters with requires_grad=False get None."""
    from dm_nerf_amd.autograd import FieldFn
    _, _, net, _, _ = nets(13, 3, 4)
    o, d, _, _ = cases.path_rays(16, seed=5)
    z = dm.z_val_sample(16, cases.NEAR, cases.FAR, 64)
    net.rgb_linear.weight.requires_grad_(False)
    raw = FieldFn.apply(G(o), G(d), z, net, "fp32", *net.parameters())
    raw[..., 4:].sum().backward()
    assert net.rgb_linear.weight.grad is None
    for k, p in net.named_parameters():
        if k.startswith("mlps.") or k.startswith("rgb_feature") or k.startswith("density"):
            assert float(p.grad.abs().max()) == 0.0, k
        elif k.startswith("ins_"):
            assert float(p.grad.abs().max()) > 0.0, k


# ---- emptiness penalizer (row N2) ---------------------------------------------------------------------------
def test_emptiness_penalizer_matches_reference_golden(golden):
    """Loss and d loss / d raw against the reference's ins_penalizer + autograd (G10): Gaussian band around the
    rendered depth, saturated logits, depth in front of / behind all samples, C = 98."""
    g = golden("g10_penalizer.npz")
    for tag, (n, s, c, tol, width, seed) in cases.PENALIZER_CASES.items():
        raw, z, depth, d = cases.penalizer_case(n, s, c, seed)
        assert cases.digest(raw, z, depth, d) == str(g[f"{tag}_digest"])
        raw_g = G(raw).requires_grad_(True)
        loss = dm.ins_penalizer(raw_g, G(z), G(depth), G(d), SimpleNamespace(tolerance=tol, deta_w=width))
        assert loss.shape == () and abs(float(loss) - float(g[f"{tag}_loss"])) < 2e-6 * abs(float(g[f"{tag}_loss"]))
        (2.5 * loss).backward()
        assert rel_err(H(raw_g.grad), 2.5 * g[f"{tag}_grad"]) < 1e-5, tag
        assert float(raw_g.grad[..., :4].abs().max()) == 0.0
    # the same through the trainer's hook on a stride-0 depth lattice
    zl = dm.z_val_sample(6, cases.NEAR, cases.FAR, 64)
    raw, _, depth, d = cases.penalizer_case(6, 64, 18, 305)
    a = dm.emptiness_penalizer(G(raw), zl, G(depth)[:, None], G(d), 0.05, 0.05)
    b = oracle.emptiness_penalizer(torch.from_numpy(raw), zl.cpu().contiguous(), torch.from_numpy(depth),
                                   torch.from_numpy(d), 0.05, 0.05)
    assert abs(float(a) - float(b)) < 2e-6 * abs(float(b))


# ---- instance-matching loss on class statistics (row N4) ---------------------------------------------------
def test_ins_criterion_matches_reference_golden(golden):
    """The four loss terms and d total / d pred against the reference's ins_criterion (G11): 5 of 13 labels present,
    all present, ins_num = 93, a single class; saturated predictions."""
    g = golden("g11_criterion.npz")
    for tag, (n, ins_num, present, seed) in cases.CRITERION_CASES.items():
        pred, labels = cases.criterion_case(n, ins_num, present, seed)
        assert cases.digest(pred, labels) == str(g[f"{tag}_digest"])
        pred_g = G(pred).requires_grad_(True)
        total, ce, inv, siou = dm.ins_criterion(pred_g, torch.from_numpy(labels).to(DEV), ins_num)
        for name, got in (("total", total), ("ce", ce), ("inv", inv), ("siou", siou)):
            want = float(g[f"{tag}_{name}"])
            assert abs(float(got.detach()) - want) <= 2e-6 * max(abs(want), 1e-3), (tag, name, float(got.detach()), want)
        total.backward()
        got, want = H(pred_g.grad), g[f"{tag}_grad"]
        np.testing.assert_allclose(got, want, rtol=2e-5, atol=1e-7 * np.abs(want).max())
        np.testing.assert_allclose(got[2:], want[2:], rtol=2e-5, atol=2e-6 * np.abs(want[2:]).max())   # unsaturated rows


def test_fused_loss_head_equals_the_tensor_operation_form():
    """dm.ins_criterion runs as five kernels around the host's Hungarian step; ``fused=False`` is the same arithmetic in
    tensor operations (the form pinned by G11 since round 1).  All four terms and the gradient of EACH of them (G11 holds
    the gradient of the total only), for every criterion case; labels the fused kernels do not take fall back."""
    for tag, (n, ins_num, present, seed) in cases.CRITERION_CASES.items():
        pred, labels = cases.criterion_case(n, ins_num, present, seed)
        lab = torch.from_numpy(labels).to(DEV)
        for term in range(4):
            grads, vals = [], []
            for fused in (True, False):
                p = G(pred).requires_grad_(True)
                out = dm.ins_criterion(p, lab, ins_num, fused=fused)
                vals.append([float(o.detach()) for o in out])
                if out[term].requires_grad:
                    out[term].backward()
                    grads.append(H(p.grad))
            for a, b in zip(*vals):
                assert abs(a - b) <= 2e-6 * max(abs(b), 1e-3), (tag, vals)
            if len(grads) == 2:
                np.testing.assert_allclose(grads[0], grads[1], rtol=2e-5, atol=2e-6 * np.abs(grads[1]).max(), err_msg=f"{tag} term {term}")
    # two predictions behind one wait for the device: the same values and gradients as two calls
    tag = "a"
    n, ins_num, present, seed = cases.CRITERION_CASES[tag]
    pred, labels = cases.criterion_case(n, ins_num, present, seed)
    pred2 = np.ascontiguousarray(pred[::-1])
    lab = torch.from_numpy(labels).to(DEV)
    pa, pb = G(pred).requires_grad_(True), G(pred2).requires_grad_(True)
    both = dm.ins_criterion_many([pa, pb], lab, ins_num)
    (both[0][0] + 2.0 * both[1][3]).backward()
    qa, qb = G(pred).requires_grad_(True), G(pred2).requires_grad_(True)
    one_a, one_b = dm.ins_criterion(qa, lab, ins_num), dm.ins_criterion(qb, lab, ins_num)
    (one_a[0] + 2.0 * one_b[3]).backward()
    for got, want in zip(both[0] + both[1], one_a + one_b):
        assert float(got.detach()) == float(want.detach())
    assert torch.equal(pa.grad, qa.grad) and torch.equal(pb.grad, qb.grad)
    # training-size batches: several ray chunks per class in the statistics kernel, label histogram over many passes
    for n, ins_num, present in ((5000, 13, (0, 1, 2, 5, 7, 11, 12)), (4096, 93, tuple(range(0, 93, 3))), (1237, 40, (3, 39))):
        rng = np.random.default_rng(n)
        labels = rng.choice(np.asarray(present), size=n).astype(np.int64)
        pred = rng.uniform(0.01, 0.99, size=(n, ins_num)).astype(np.float32)
        lab = torch.from_numpy(labels).to(DEV)
        vals, grads = [], []
        for fused in (True, False):
            p = G(pred).requires_grad_(True)
            out = dm.ins_criterion(p, lab, ins_num, fused=fused)
            out[0].backward()
            vals.append([float(o.detach()) for o in out])
            grads.append(H(p.grad))
        for a, b in zip(*vals):
            assert abs(a - b) <= 2e-6 * max(abs(b), 1e-3), (n, vals)
        np.testing.assert_allclose(grads[0], grads[1], rtol=2e-5, atol=2e-6 * np.abs(grads[1]).max())
    # labels beyond the fused kernels' table (>= 1024): same values through the fallback
    pred, labels = cases.criterion_case(*cases.CRITERION_CASES[next(iter(cases.CRITERION_CASES))])
    far = torch.from_numpy(labels).to(DEV) * 1000
    a = dm.ins_criterion(G(pred), far, cases.CRITERION_CASES[next(iter(cases.CRITERION_CASES))][1])
    b = dm.ins_criterion(G(pred), far, cases.CRITERION_CASES[next(iter(cases.CRITERION_CASES))][1], fused=False)
    assert [float(x) for x in a] == [float(x) for x in b]


def test_device_side_assignment_is_scipys_assignment():
    """``assign="device"``: the Hungarian step as a kernel (scipy's shortest-augmenting-path algorithm in fp64) gives the
    loss terms and gradients of the host form bit for bit -- every G11 case, training-size batches, and predictions with
    duplicated and constant columns, where the choice among equally good columns is scipy's scan order."""
    sets = [cases.criterion_case(n, k, present, seed) + (k,) for n, k, present, seed in cases.CRITERION_CASES.values()]
    rng = np.random.default_rng(77)
    for n, k, present in ((4096, 13, (0, 1, 2, 5, 7, 11, 12)), (3000, 93, tuple(range(0, 93, 2))), (777, 40, tuple(range(40)))):
        sets.append((rng.uniform(0.01, 0.99, size=(n, k)).astype(np.float32),
                     rng.choice(np.asarray(present), size=n).astype(np.int64), k))
    pred, labels, k = sets[0]
    dup = pred.copy(); dup[:, 1] = dup[:, 0]; dup[:, 5] = dup[:, 0]; dup[:, 9] = dup[:, 8]      # equal cost columns
    sets.append((dup, labels, k))
    sets.append((np.full_like(pred, 0.5), labels, k))                                          # a constant cost matrix
    for pred, labels, k in sets:
        lab = torch.from_numpy(labels).to(DEV)
        got = []
        for assign in ("host", "device"):
            p = G(pred).requires_grad_(True)
            out = dm.ins_criterion(p, lab, k, assign=assign)
            (out[0] + 0.5 * out[3]).backward()
            got.append(([float(o.detach()) for o in out], p.grad.clone()))
        assert got[0][0] == got[1][0], (got[0][0], got[1][0])
        assert torch.equal(got[0][1], got[1][1])
    # labels the kernels do not take: NaN instead of the host form's fallback
    bad = dm.ins_criterion(G(sets[0][0]), torch.from_numpy(sets[0][1]).to(DEV) * 1000, sets[0][2], assign="device")
    assert all(np.isnan(float(t)) for t in bad)


# ---- ray-sharded training is batch-exact ----------------------------------------------
class _TwoShardReducer:
    """Single-process stand-in for parallel.ShardReducer: pass 1 records what a shard would contribute, pass 2 returns
    own + the other shard's recording (detached: the other rank's backward is its own business)."""

    def __init__(self, known=None):
        self.own, self.other, self.calls, self.known = [], None, 0, known or {}

    def __call__(self, tag, t):
        if self.other is None:
            self.own.append((tag, t.detach().clone()))
            # the label set shapes the statistics (one row per label present anywhere): the recording pass must
            # already see the whole batch's
            if tag not in self.known:
                return t
            k = self.known[tag].to(t.device)
            if k.numel() + 1 == t.numel():          # the kernels' histogram carries the out-of-range flag as a last entry
                k = torch.cat([k, k.new_zeros(1)])
            return k
        tag2, rec = self.other[self.calls]
        assert tag2 == tag
        self.calls += 1
        return t + rec


@pytest.mark.parametrize("fused", [True, False])
def test_sharded_loss_heads_reproduce_the_unsharded_loss_and_gradient(fused):
    """(With the instance head as kernels -- histogram, statistics and ray count summed between the launches -- and in
    its tensor-operation form.)  ins_criterion and ins_penalizer on two unequal ray shards with their batch statistics summed (what
    parallel.ShardReducer's all-reduce does): every shard computes the unsharded loss VALUE, and the shards' gradients
    together are the unsharded gradient -- the Hungarian matching sees the whole batch's cost matrices and the label
    set of the whole batch (one label occurs in shard B only), the penalizer the whole batch's mask sums."""
    n, ins_num, cut = 96, 13, 41
    pred, labels = cases.criterion_case(n, ins_num, (0, 2, 3, 7, 12), 311)
    labels[:cut][labels[:cut] == 12] = 7                                     # label 12 is absent from shard A
    assert 12 in labels[cut:] and 12 not in labels[:cut]
    raw, z, depth, d = cases.penalizer_case(n, 64, 18, 301)
    pargs = SimpleNamespace(tolerance=0.3, deta_w=0.2)
    lab = torch.from_numpy(labels).to(DEV)

    def losses(sl, reduce):
        p = G(pred[sl]).requires_grad_(True)
        r = G(raw[sl]).requires_grad_(True)
        ins = dm.ins_criterion(p, lab[sl], ins_num, reduce=reduce, fused=fused)[0]
        pen = dm.ins_penalizer(r, G(z[sl]), G(depth[sl]), G(d[sl]), pargs, reduce=reduce)
        (ins + pen).backward()
        return float(ins.detach()), float(pen.detach()), H(p.grad), H(r.grad)

    whole = losses(slice(0, n), None)
    shards = (slice(0, cut), slice(cut, n))
    rec = []
    known = {"ins_label_count": torch.bincount(torch.from_numpy(labels), minlength=ins_num + 1).float(),
             "ins_ray_count": torch.tensor([float(n)])}
    for sl in shards:                                                        # pass 1: what each shard contributes
        r = _TwoShardReducer(known)
        losses(sl, r)
        rec.append(r.own)
    got = []
    for k, sl in enumerate(shards):                                          # pass 2: with the other shard's sums
        r = _TwoShardReducer()
        r.other = rec[1 - k]
        got.append(losses(sl, r))
        assert r.calls == len(rec[1 - k])
    for k in range(2):                                                       # the loss values are the whole batch's
        assert abs(got[k][0] - whole[0]) < 2e-6 * abs(whole[0]), (got[k][0], whole[0])
        assert abs(got[k][1] - whole[1]) < 2e-6 * abs(whole[1]), (got[k][1], whole[1])
    g_pred = np.concatenate([got[0][2], got[1][2]], 0)
    g_raw = np.concatenate([got[0][3], got[1][3]], 0)
    assert rel_err(g_pred, whole[2]) < 2e-6 and rel_err(g_raw, whole[3]) < 2e-6
    # and without the reduction the shards optimise something else (what round 1 did)
    alone = losses(shards[0], None)
    assert abs(alone[0] - whole[0]) > 1e-3 * abs(whole[0])


@pytest.mark.parametrize("fused", [True, False])
def test_sharded_instance_loss_stops_every_shard_on_a_label_out_of_range(fused):
    """A label outside [0, ins_num] on ONE shard: the flag travels with the label histogram through the reduction, so the
    clean shard raises too (on a real process group every rank leaves before the next collective; a rank-local check
    would leave the others waiting in it)."""
    n, ins_num = 64, 13
    pred, labels = cases.criterion_case(n, ins_num, (0, 2, 3, 7, 12), 317)
    bad = labels.copy()
    bad[5] = ins_num + 4
    lab_bad, lab_ok = torch.from_numpy(bad[:32]).to(DEV), torch.from_numpy(labels[32:]).to(DEV)
    rec = _TwoShardReducer()                                                 # recording pass: the shard's own contribution
    with pytest.raises(ValueError):                                          # the shard that holds the label
        dm.ins_criterion(G(pred[:32]), lab_bad, ins_num, reduce=rec, fused=fused)
    assert rec.own[0][0] == "ins_label_count" and float(rec.own[0][1][-1]) == 1.0
    clean = _TwoShardReducer()
    clean.other = rec.own                                                    # the clean shard sees the summed flag
    with pytest.raises(ValueError):
        dm.ins_criterion(G(pred[32:]), lab_ok, ins_num, reduce=clean, fused=fused)
    # and both shards issued the same reductions before they stopped (the tensor-operation form reads the flag right
    # behind the first one, the kernels' form with its one read-back behind the third)
    assert clean.calls == len(rec.own) == (3 if fused else 1)


def test_device_side_assignment_surfaces_a_label_out_of_range_at_the_loss_read_back():
    """Ray-sharded training with assign="device" never waits for the loss head's status words: a label outside
    [0, ins_num] (the sharded path's label bins) gives non-finite instance terms and zero gradients from that head
    instead of the ValueError the host form raises at the call.  The step's loss read-back
    (trainer.train_step(report=True)) is the first point that waits for the device: it raises."""
    from dm_nerf_amd import trainer
    n, ins_num = 64, 13
    pos, view, coarse, fine, args = nets(ins_num, 0, 1)
    coarse.train(); fine.train()
    o, d, _, _ = cases.path_rays(n, seed=29)
    targs = SimpleNamespace(**{**vars(args), "perturb": 1.0, "is_train": True})
    _, labels = cases.criterion_case(n, ins_num, (0, 2, 3, 7, 12), 317)
    bad = labels.copy()
    bad[5] = ins_num + 4
    z = dm.z_val_sample(n, cases.NEAR, cases.FAR, 64)
    target = G(np.random.default_rng(3).uniform(0, 1, (n, 3)).astype(np.float32))
    opt = trainer.make_optimizer(coarse, fine)
    one_shard = lambda tag, t: t                                      # the sharded code path on a world of one
    step = lambda lab, report: trainer.train_step(
        0, G(np.stack([o, d], 0)), target, (pos, view, coarse, fine, targs), z, opt,
        ins_loss=trainer.instance_loss(torch.from_numpy(lab).to(DEV), ins_num, assign="device", reduce=one_shard),
        report=report)
    good = step(labels, True)
    assert math.isfinite(good["ins_fine"]) and math.isfinite(good["total"])
    quiet = step(bad, False)                                          # no read-back: nothing can raise here
    assert not math.isfinite(float(quiet["ins_fine"]))
    with pytest.raises(ValueError, match="label outside"):
        step(bad, True)
    with pytest.raises(ValueError):                                          # the host form raises at the call
        trainer.train_step(0, G(np.stack([o, d], 0)), target, (pos, view, coarse, fine, targs), z, opt,
                           ins_loss=trainer.instance_loss(torch.from_numpy(bad).to(DEV), ins_num, assign="host",
                                                          reduce=one_shard))


def test_graph_captured_training_step_equals_the_eager_steps():
    """trainer.GraphedTrainStep: the whole step (both forwards, loss heads with the device-side Hungarian step, hand-written
    backward, Adam, learning-rate decay) replayed from one hipGraph against the same number of eager train_step() calls
    from the same start: same loss terms and weights (Adam's capturable form differs from the default one by rounding),
    the decayed learning rate of train_dmsr.py:68-72, inputs read from the static buffers at every replay."""
    from dm_nerf_amd import trainer
    n, ins_num, lrate, decay = 256, 13, 5e-4, 1
    o, d, _, _ = cases.path_rays(n, seed=37)
    rays = G(np.stack([o, d], 0))
    rng = np.random.default_rng(5)
    target = G(rng.uniform(0, 1, (n, 3)).astype(np.float32))
    labels = torch.from_numpy(rng.integers(0, ins_num, n)).to(DEV)
    z = dm.z_val_sample(n, cases.NEAR, cases.FAR, 64)

    def fresh():
        pos, view, coarse, fine, args = nets(ins_num, 0, 1)
        coarse.train(); fine.train()
        targs = SimpleNamespace(**{**vars(args), "perturb": 0.0, "is_train": True, "tolerance": 0.05, "deta_w": 0.05})
        return (pos, view, coarse, fine, targs)

    def heads(nerf):
        return dict(ins_loss=trainer.instance_loss(labels, ins_num, assign="device"),
                    penalizer=lambda raw, zz, depth: dm.ins_penalizer(raw, zz, depth, rays[1], nerf[4]))

    eager = fresh()
    opt = trainer.make_optimizer(eager[2], eager[3], lrate)
    for i in range(4):
        last = trainer.train_step(i, rays, target, eager, z, opt, lrate=lrate, lrate_decay=decay, report=False, **heads(eager))
    graphed = fresh()
    step = trainer.GraphedTrainStep(rays, target, graphed, z, lrate=lrate, lrate_decay=decay, warmup=2, **heads(graphed))
    for _ in range(2):                                                       # 2 warm-up steps + the captured one... = 3; one more
        terms = step()                                                       # replay: steps 3 and 4 (the capture itself runs nothing)
    assert float(step.iteration) == 4.0
    assert abs(float(step.lr) - trainer.decayed_lrate(3, lrate, decay)) < 1e-9       # set after step 3 for step 4, as :68-72
    # (four steps: the two Adam forms differ by rounding from the first update on, and behind the resampler a rounding
    # moves single depths -- observed 8e-5 on the emptiness term, 1e-5 on the others)
    for k in last:
        assert abs(float(terms[k]) - float(last[k])) < 1e-3 * max(1.0, abs(float(last[k]))), (k, float(terms[k]), float(last[k]))
    # Adam's update is m / (sqrt(v) + eps): for an entry whose gradient is within rounding of zero, rounding decides the
    # update's size -- weights are compared against what four steps can move them at all (4 x lrate = 2e-3; observed 1.5e-5)
    for (k, a), b in zip(eager[3].named_parameters(), graphed[3].parameters()):
        assert float((a - b).abs().max()) < 0.05 * 4 * lrate, k
    # the replay reads its inputs from the static buffers: new targets change the colour loss
    before = float(step()["rgb_fine"])
    step.target.copy_(1.0 - step.target)
    assert abs(float(step()["rgb_fine"]) - before) > 1e-3


def test_grad_bucket_receives_the_weight_gradients_in_place():
    """parallel.GradBucket: the weight-gradient kernel of both networks writes straight into the persistent flat
    buffer, p.grad are views of it after backward (nothing to concatenate or copy back), and the values equal a
    bucket-less backward."""
    from dm_nerf_amd.parallel import GradBucket
    pos, view, coarse, fine, args = nets(13, 0, 1)
    coarse.train(); fine.train()
    o, d, t_rand, u = cases.path_rays(32, seed=23)
    args = SimpleNamespace(**{**vars(args), "perturb": 1.0, "is_train": True, "t_rand": G(t_rand), "u": G(u)})
    z = dm.z_val_sample(32, cases.NEAR, cases.FAR, 64)

    def backward():
        out = dm.dm_nerf(G(np.stack([o, d], 0)), pos, view, coarse, fine, z, args)
        (out["rgb_fine"].sum() + out["rgb_coarse"].sum() + out["ins_fine"].sum()).backward()

    backward()
    plain = torch.cat([p.grad.reshape(-1) for m in (coarse, fine) for p in m.parameters()]).clone()
    bucket = GradBucket([coarse, fine])
    for step in range(2):                                                    # buffer reuse across steps
        for m in (coarse, fine):
            for p in m.parameters():
                p.grad = None
        bucket.flat.fill_(float("nan"))                                      # every entry must be overwritten
        backward()
        in_place = [p.grad.data_ptr() == v.data_ptr() for p, v in zip(bucket.params, bucket.views)]
        assert all(in_place), f"{sum(in_place)} of {len(in_place)} gradients live in the bucket"
        np.testing.assert_array_equal(H(bucket.flat), H(plain))
    with pytest.raises(RuntimeError, match="save record is released"):
        raw = dm.autograd.FieldFn.apply(G(o), G(d), z, coarse, "fp32", *coarse.parameters())
        raw.sum().backward(retain_graph=True)
        raw.sum().backward()


def test_grad_bucket_with_one_model_applied_twice_in_one_graph():
    """Two ray batches through the SAME networks, summed into one loss, under a GradBucket: the engine sums the two
    field nodes' parameter gradients only after both backward nodes have run, so the second node must not be handed the
    bucket's buffer while the first node's result lives in it (it would be overwritten in place: g2 + g2).  The
    gradients must equal the bucket-less ones, and the lending must end with the backward pass (the next step writes
    in place again)."""
    from dm_nerf_amd.parallel import GradBucket
    pos, view, coarse, fine, args = nets(13, 0, 1)
    coarse.train(); fine.train()
    o, d, t_rand, u = cases.path_rays(32, seed=29)
    z = dm.z_val_sample(16, cases.NEAR, cases.FAR, 64)

    def backward():
        total = 0.0
        for k, sl in enumerate((slice(0, 16), slice(16, 32))):
            a = SimpleNamespace(**{**vars(args), "perturb": 1.0, "is_train": True, "t_rand": G(t_rand[sl]), "u": G(u[sl])})
            out = dm.dm_nerf(G(np.stack([o[sl], d[sl]], 0)), pos, view, coarse, fine, z, a)
            total = total + (1.0 + k) * (out["rgb_fine"].sum() + out["rgb_coarse"].sum() + out["ins_fine"].sum())
        total.backward()

    def clear():
        for m in (coarse, fine):
            for p in m.parameters():
                p.grad = None

    backward()
    plain = torch.cat([p.grad.reshape(-1) for m in (coarse, fine) for p in m.parameters()]).clone()
    clear()
    bucket = GradBucket([coarse, fine])
    for step in range(2):
        clear()
        bucket.flat.fill_(float("nan"))
        backward()
        assert not coarse._grad_lent and not fine._grad_lent             # returned at the end of the backward pass
        bucket._collect()                                                # what allreduce() does first
        assert rel_err(H(bucket.flat), H(plain)) < 1e-6, step
    # and a single application still lands in place afterwards
    clear()
    a = SimpleNamespace(**{**vars(args), "perturb": 1.0, "is_train": True, "t_rand": G(t_rand[:16]), "u": G(u[:16])})
    out = dm.dm_nerf(G(np.stack([o[:16], d[:16]], 0)), pos, view, coarse, fine, z, a)
    (out["rgb_fine"].sum() + out["rgb_coarse"].sum()).backward()
    assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(bucket.params, bucket.views))


def test_assignment_failure_is_per_call_and_leaves_the_label_cache_usable():
    """assign="device" on a cost matrix without a finite matching (a NaN prediction): that call returns NaN terms; the
    label classes cached for the labels tensor are untouched, so the next call on the same labels is the normal one."""
    n, ins_num = 64, 13
    pred, labels = cases.criterion_case(n, ins_num, (0, 2, 3, 7, 12), 313)
    lab = torch.from_numpy(labels).to(DEV)
    good = dm.ins_criterion(G(pred), lab, ins_num, assign="device")
    bad_pred = pred.copy()
    bad_pred[:] = np.nan                                                     # no finite path cost anywhere
    bad = dm.ins_criterion(G(bad_pred), lab, ins_num, assign="device")
    assert not np.isfinite(float(bad[0]))
    again = dm.ins_criterion(G(pred), lab, ins_num, assign="device")
    host = dm.ins_criterion(G(pred), lab, ins_num, assign="host")
    for a, b, c in zip(good, again, host):
        assert float(a) == float(b) == float(c)


# ---- training ray selectors (row N1) --------------------------------------------------------------------
def test_ray_selectors_match_reference_golden(golden):
    g = golden("g9_selectors.npz")
    c = cases.selector_case()
    tc, ti, rays = dm.get_select_full(G(c["rgb"]), c["pose"], c["K"], G(c["label"]), c["n_train"],
                                      selected_index=c["full_pick"])
    np.testing.assert_array_equal(H(tc), g["full_target_c"])
    np.testing.assert_array_equal(H(ti), g["full_target_i"])
    np.testing.assert_array_equal(H(rays[0]), g["full_rays"][0])
    np.testing.assert_allclose(H(rays), g["full_rays"], rtol=0, atol=1e-6)
    tc, ti, rays, n_ins = dm.get_select_crop(G(c["rgb"]), c["pose"], c["K"], G(c["label"]), c["ins_index"],
                                             c["crop_mask"], c["n_train"], labeled_pick=c["labeled_pick"],
                                             unlabeled_pick=c["unlabeled_pick"])
    assert n_ins == int(g["crop_n_ins"]) and rays.shape == (2, c["n_train"], 3)
    np.testing.assert_array_equal(H(tc), g["crop_target_c"])
    np.testing.assert_array_equal(H(ti), g["crop_target_i"])
    np.testing.assert_allclose(H(rays), g["crop_rays"], rtol=0, atol=1e-6)
    # own draws: distinct pixels, right sizes, rays equal to the full-image generator at those pixels
    tc, ti, rays = dm.get_select_full(G(c["rgb"]), c["pose"], c["K"], G(c["label"]), 64)
    assert tc.shape == (64, 3) and ti.shape == (64,) and rays.shape == (2, 64, 3)
    ro, rd = dm.get_rays_k(c["H"], c["W"], c["K"], c["pose"])
    flat = rd.reshape(-1, 3)
    match = (flat[None, :, :] == rays[1][:, None, :]).all(-1)
    assert bool(match.any(1).all()) and int(match.any(0).sum()) == 64


# ---- image loop with a density-only coarse pass -------------------------------------------------------------
@pytest.mark.parametrize("precision", ["bf16x3", "fp16x3", "bf16x6"])
def test_image_only_coarse_pass_gives_bit_identical_images(precision):
    """render_image(image_only=True): the coarse network stops behind its density head; sigma, the resampled depths and
    both returned images must not change by a bit."""
    pos, view, coarse, fine, args = nets(13, 0, 1)
    args.precision = precision
    Hh, W = 24, 20
    hwk = [Hh, W, cases.study_intrinsics(Hh, W)]
    rgb, ins = dm.render_image(pos, view, coarse, fine, cases.study_c2w(), hwk, args, chunk=200)
    rgb2, ins2 = dm.render_image(pos, view, coarse, fine, cases.study_c2w(), hwk, args, chunk=200, image_only=True)
    np.testing.assert_array_equal(H(rgb), H(rgb2))
    np.testing.assert_array_equal(H(ins), H(ins2))
    o, d, _, _ = cases.path_rays(70, seed=9)
    z = dm.z_val_sample(70, cases.NEAR, cases.FAR, 64)
    full = ops.field_rays(G(o), G(d), z, coarse.blob(precision), 13, precision)
    part = ops.field_rays(G(o), G(d), z, coarse.blob(precision), 13, precision, sigma_only=True)
    np.testing.assert_array_equal(H(full[..., 3]), H(part[..., 3]))
    assert float(part[..., :3].abs().max()) == 0.0 and float(part[..., 4:].abs().max()) == 0.0
    with pytest.raises(ValueError):
        ops.field_rays(G(o), G(d), z, coarse.blob("fp32"), 13, "fp32", sigma_only=True)


# ---- field + compositing (+ resampling) fused in one kernel: the image loop's path in the split precisions ---------
@pytest.mark.parametrize("precision", ["bf16x3", "fp16x3", "bf16x6"])
@pytest.mark.parametrize("ins_num,n_rays,n_samples", [(13, 37, 64), (13, 6, 192), (93, 9, 64), (93, 4, 192), (127, 3, 128),
                                                      (0, 2, 64)])
def test_fused_field_composite_is_bit_identical_to_the_chain(precision, ins_num, n_rays, n_samples):
    """dmn_field_composite_x3 (field evaluation, compositing and resampling in one kernel; raw rows parked in LDS) against
    dmn_field_fwd_rays_x3 -> dmn_composite_fwd -> dmn_resample_merge: every output, bit for bit.  Odd ray counts (a
    workgroup owns two rays), 1 / 2 / 3 passes of 64 samples, narrow and wide rows (C = 18 / 98 / 132 / 5), recorded
    draws and the det = True lattice, and the density-only mode."""
    precision = ops.save_blob_name(precision)       # bf16x6: the chain on the kernel family the fused kernel is built from
    rng = np.random.default_rng(1000 * ins_num + n_rays)
    params = oracle.make_params(ins_num, 7)
    model = dm.DM_NeRF(8, 256, 63, 27, [4], ins_num).to(DEV)
    model.load_state_dict(params)
    blob = model.blob(precision)
    o = G(rng.uniform(-1, 1, size=(n_rays, 3)).astype(np.float32))
    d = G(rng.standard_normal((n_rays, 3)).astype(np.float32))
    z = G(np.sort(rng.uniform(cases.NEAR, cases.FAR, size=(n_rays, n_samples)).astype(np.float32), -1))
    u = G(rng.uniform(0, 1, size=(n_rays, 128)).astype(np.float32))
    raw = ops.field_rays(o, d, z, blob, ins_num, precision)
    rgb, w, depth, ins = ops.composite(raw, z, d)
    for draws in (u, None):
        z_fine = ops.resample_merge(z, w, 128, draws)
        got = ops.field_composite(o, d, z, blob, ins_num, precision, weights=True, n_new=128, u=draws)
        for name, a, b in (("rgb", got[0], rgb), ("depth", got[1], depth), ("ins", got[2], ins), ("weights", got[3], w),
                           ("z_fine", got[4], z_fine)):
            np.testing.assert_array_equal(H(a), H(b), err_msg=f"{name} (draws given: {draws is not None})")
    # shared lattice row (stride 0), maps only
    lat = z[:1].expand(n_rays, n_samples)
    raw_l = ops.field_rays(o, d, lat, blob, ins_num, precision)
    rgb_l, _, depth_l, ins_l = ops.composite(raw_l, lat, d)
    got = ops.field_composite(o, d, lat, blob, ins_num, precision)
    assert got[3] is None and got[4] is None
    for a, b in ((got[0], rgb_l), (got[1], depth_l), (got[2], ins_l)):
        np.testing.assert_array_equal(H(a), H(b))
    # density-only coarse pass: weights and resampled depths unchanged
    got = ops.field_composite(o, d, z, blob, ins_num, precision, sigma_only=True, maps=False, weights=True, n_new=128, u=u)
    assert got[0] is None
    np.testing.assert_array_equal(H(got[3]), H(w))
    np.testing.assert_array_equal(H(got[4]), H(ops.resample_merge(z, w, 128, u)))
    assert ops.take_status(DEV) == 0
    with pytest.raises(dm.HipLibraryError):                                  # 64-sample passes only
        ops.field_composite(o, d, z[:, :50].contiguous(), blob, ins_num, precision)


@pytest.mark.parametrize("precision", ["bf16x3", "fp16x3", "bf16x6"])
def test_image_loop_fused_path_equals_the_chained_path(precision):
    """render_image in a split precision runs ops.render_fused; args.fused = False runs dm_nerf()'s chain: same bits,
    ragged last chunk with an odd ray count, with and without the density-only coarse pass, Replica-wide rows.  (bf16x6:
    the fused kernel is built from the 32x32x16 network, so the chain it is compared with runs "bf16x6_32"; the default
    bf16x6 inference kernel sums in another order.)"""
    precision = ops.save_blob_name(precision)
    for ins_num, (Hh, W), near, far in ((13, (21, 25), cases.NEAR, cases.FAR), (93, (9, 15), 0.0, 6.5)):
        pos, view, coarse, fine, args = nets(ins_num, 3, 5)
        args.precision, args.near, args.far = precision, near, far
        hwk = [Hh, W, cases.study_intrinsics(Hh, W)]
        c2w = cases.study_c2w()
        plain = SimpleNamespace(**{**vars(args), "fused": False})
        forced = SimpleNamespace(**{**vars(args), "fused": True})
        a = dm.render_image(pos, view, coarse, fine, c2w, hwk, forced, chunk=200)
        b = dm.render_image(pos, view, coarse, fine, c2w, hwk, plain, chunk=200)
        c = dm.render_image(pos, view, coarse, fine, c2w, hwk, forced, chunk=200, image_only=True)
        auto = dm.render_image(pos, view, coarse, fine, c2w, hwk, args, chunk=200)        # the default rule, either way
        np.testing.assert_array_equal(H(auto[0]), H(b[0]))
        for x, y in ((a[0], b[0]), (a[1], b[1]), (c[0], b[0]), (c[1], b[1])):
            np.testing.assert_array_equal(H(x), H(y))
    out = ops.render_fused(*[t.reshape(-1, 3) for t in dm.get_rays_k(Hh, W, hwk[2], c2w)],
                           dm.z_val_sample(Hh * W, near, far, 64), coarse.blob(precision), fine.blob(precision), ins_num,
                           precision, 128, coarse_maps=True)
    with torch.no_grad():
        rays = torch.stack([t.reshape(-1, 3) for t in dm.get_rays_k(Hh, W, hwk[2], c2w)], 0)
        want = dm.dm_nerf(rays, pos, view, coarse, fine, dm.z_val_sample(Hh * W, near, far, 64), plain)
    for k in ("rgb_fine", "depth_fine", "ins_fine", "z_vals_fine", "rgb_coarse", "depth_coarse", "ins_coarse"):
        np.testing.assert_array_equal(H(out[k]), H(want[k]), err_msg=k)


# ---- object manipulation (row N3), pinned by the reference's own functions (G12) ------------------------------
@pytest.mark.parametrize("tag", sorted(cases.EXCHANGE_CASES))
def test_exchanger_is_bit_equal_to_the_reference(golden, tag):
    """exchanger() against networks/manipulator.py:18-83 run on the same inputs: two / one / three moved objects
    (one label that never occurs), C = 18 / 98 / 11, every branch (occluders, fillings, take, wipe, keep), saturated
    logits so that the sigmoid's ties matter.  Labels, and every byte of the edited tensor."""
    g = golden("g12_manipulator.npz")
    n, s, c, moves, seed = cases.EXCHANGE_CASES[tag]
    ori_raw, ori_pred, tar_raws, tar_preds = cases.exchange_case(n, s, c, moves, seed)
    assert str(g[f"ex_{tag}_digest"]) == cases.digest(ori_raw, ori_pred, *tar_raws, *tar_preds)
    got = G(ori_raw)
    out_raw, _, ol, tl = dm.exchanger(got, [G(t) for t in tar_raws], G(ori_pred), [G(t) for t in tar_preds], list(moves))
    assert out_raw.data_ptr() == got.data_ptr()                 # edited in place, like the reference
    np.testing.assert_array_equal(H(ol), g[f"ex_{tag}_ori_label"])
    np.testing.assert_array_equal(H(tl), g[f"ex_{tag}_tar_label"])
    assert ol.dtype == torch.int64 and tl.dtype == torch.int64
    assert cases.digest(H(out_raw)) == str(g[f"ex_{tag}_out_digest"])
    src = g[f"ex_{tag}_src"]
    np.testing.assert_array_equal(H(out_raw)[src == -1], 0.0)
    for k, t in enumerate(tar_raws):
        np.testing.assert_array_equal(H(out_raw)[src == 1 + k], t[src == 1 + k])


def test_manipulator_render_and_nerf_match_the_reference(golden):
    g = golden("g12_manipulator.npz")
    for tag, (n, s, c, seed) in {"a": (32, 64, 18, 141), "b": (16, 192, 18, 142), "c": (8, 64, 98, 143),
                                 "d": (8, 100, 11, 144)}.items():
        raw, z, d = cases.composite_case(n, s, c, seed)
        got = dm.manipulator_render(G(raw), G(z), G(d))
        for name, val in zip(("rgb", "w", "depth", "ins"), got):
            assert val.shape == g[f"mr_{tag}_{name}"].shape
            assert rel_err(H(val), g[f"mr_{tag}_{name}"]) < 5e-6, (tag, name)
    o, d, _, _ = cases.manipulator_case()
    pos, view, coarse, fine, args = nets(13, 0, 1)
    coarse.load_state_dict(cases.manipulator_params(13, *cases.MANIPULATOR_NETS[0]))
    fine.load_state_dict(cases.manipulator_params(13, *cases.MANIPULATOR_NETS[1]))
    rays = G(np.stack([o, d], 0))[:, :6]
    with torch.no_grad():
        raw, z = dm.manipulator_nerf(rays, pos, view, coarse, cases.N_SAMPLES, cases.NEAR, cases.FAR)
        np.testing.assert_array_equal(H(z[0]), g["mn_z_lattice"])           # near*(1-t) + far*t, not z_val_sample's
        assert z.shape == (6, 64) and rel_err(H(raw), g["mn_raw"]) < 1e-5
        raw2, z2 = dm.manipulator_nerf(rays, pos, view, fine, z_vals=(z[:, ::3] * 1.01).contiguous())
        assert rel_err(H(raw2), g["mn_raw_given_z"]) < 1e-5


def test_manipulator_two_pass_rerender_matches_the_reference(golden):
    """manipulator() (networks/manipulator.py:137-205) with the reference's recorded torch.rand draws: two moved
    objects, labels 1 and 4.  Point labels of both exchanges, the merged depths of both passes and the four returned
    maps.  Like dm_nerf's fine outputs the result sits behind stochastic resamplings and, here, behind argmax label
    decisions, so a ray may leave the bound when one of its points changes label on rounding noise: the labels are held
    to >= 99.9 % agreement (observed: all equal), the maps to 1e-5 on the rays whose labels all agree (observed: all
    rays, 3e-6)."""
    g = golden("g12_manipulator.npz")
    o, d, tars, draws = cases.manipulator_case()
    pos, view, coarse, fine, args = nets(13, 0, 1)
    coarse.load_state_dict(cases.manipulator_params(13, *cases.MANIPULATOR_NETS[0]))
    fine.load_state_dict(cases.manipulator_params(13, *cases.MANIPULATOR_NETS[1]))
    rays = G(np.stack([o, d], 0))
    f_tar = [G(np.stack([a, b], 0)) for a, b in tars]
    margs = SimpleNamespace(N_samples=cases.N_SAMPLES, N_importance=cases.N_IMPORTANCE, near=cases.NEAR, far=cases.FAR,
                            target_label=list(cases.MANIPULATOR_LABELS), u_draws=[G(u) for u in draws])
    trace = {"labels": [], "z": []}
    import sys
    mod = sys.modules["dm_nerf_amd.manipulator"]            # the package attribute of that name is the function
    inner_ex, inner_nerf = mod.exchanger, mod.manipulator_nerf

    def ex_spy(*a):
        res = inner_ex(*a)
        trace["labels"].append((H(res[2]), H(res[3])))
        return res

    def nerf_spy(*a, **kw):
        res = inner_nerf(*a, **kw)
        trace["z"].append(H(res[1]))
        return res

    mod.exchanger, mod.manipulator_nerf = ex_spy, nerf_spy
    try:
        rgb, ins, tar_rgb, tar_acc = dm.manipulator(pos, view, coarse, fine, rays, f_tar, margs)
    finally:
        mod.exchanger, mod.manipulator_nerf = inner_ex, inner_nerf
    assert rgb.shape == (24, 3) and ins.shape == (24, 14) and tar_rgb.shape == (24, 3) and tar_acc.shape == (24, 14)
    # pass 1: lattice exchange (no resampling in front of it)
    np.testing.assert_array_equal(trace["labels"][0][0], g["mp_ex1_ori_label"])
    np.testing.assert_array_equal(trace["labels"][0][1], g["mp_ex1_tar_label"])
    z_first, z_final = trace["z"][1], trace["z"][-2]
    assert z_final.shape == (24, 64 + 128 * 3) and np.all(np.diff(z_final, axis=-1) >= 0)
    off1 = np.mean(np.abs(z_first - g["mp_z_full_first"]) > 1e-4 * cases.FAR)
    off2 = np.mean(np.abs(z_final - g["mp_z_final"]) > 1e-4 * cases.FAR)
    agree = [float(np.mean(trace["labels"][1][k] == g[key])) for k, key in ((0, "mp_ex2_ori_label"), (1, "mp_ex2_tar_label"))]
    ray_ok = (trace["labels"][1][0] == g["mp_ex2_ori_label"]).all(-1) & (np.abs(z_final - g["mp_z_final"]) < 1e-3).all(-1)
    errs = {}
    for got, key in ((rgb, "mp_final_rgb"), (ins, "mp_final_ins"), (tar_rgb, "mp_tar_rgb"), (tar_acc, "mp_tar_ins_accum")):
        diff = np.abs(H(got) - g[key]).reshape(24, -1).max(-1) / np.abs(g[key]).max()
        errs[key] = (float(diff[ray_ok].max()) if ray_ok.any() else 0.0, float(diff.max()), float((diff > 1e-4).mean()))
    print(f"manipulator: depths off {off1:.4f} / {off2:.4f}, pass-2 labels agree {agree}, rays with all labels equal "
          f"{ray_ok.mean():.3f}, (err on those, err on all, rays > 1e-4): {errs}")
    # observed on MI355X: every depth within 1e-4, every label equal, maps within 3.1e-6 / 1.8e-6 / 1.7e-7 / 8.4e-7
    assert off1 <= 0.002 and off2 <= 0.002
    assert min(agree) >= 0.999
    assert ray_ok.mean() >= 0.95
    for key, (e_ok, e_all, frac) in errs.items():
        assert e_ok < 1e-5, (key, e_ok)
    assert rel_err(H(tar_rgb), g["mp_tar_rgb"]) < 1e-4          # coarse pass of the last object: no resampling involved
    # own draws: deterministic under a fixed seed
    margs.u_draws = None
    torch.manual_seed(5)
    a = dm.manipulator(pos, view, coarse, fine, rays, f_tar, margs)
    torch.manual_seed(5)
    b = dm.manipulator(pos, view, coarse, fine, rays, f_tar, margs)
    for x, y in zip(a, b):
        np.testing.assert_array_equal(H(x), H(y))


@pytest.mark.parametrize("chunk", [100, 4096])
def test_graph_captured_image_loop_equals_the_eager_loop(chunk):
    """tester.GraphedImage: the whole chunk loop of an image as one hipGraph replay -- bit-identical to render_image for
    several poses from one capture (ragged last chunk; the reference's N_test and a small chunk), static output buffers."""
    Hh, W = 41, 37
    pos, view, coarse, fine, args = nets(13, 0, 1)
    hwk = [Hh, W, cases.study_intrinsics(Hh, W)]
    renderer = dm.GraphedImage(pos, view, coarse, fine, hwk, args, chunk=chunk)
    for i in (0, 3, 5):
        c2w = oracle.study_pose(i)
        rgb, ins = renderer(c2w)
        want_rgb, want_ins = dm.render_image(pos, view, coarse, fine, c2w, hwk, args, chunk=chunk)
        np.testing.assert_array_equal(H(rgb), H(want_rgb))
        np.testing.assert_array_equal(H(ins), H(want_ins))
    with pytest.raises(ValueError):
        dm.GraphedImage(pos, view, coarse, fine, hwk, SimpleNamespace(**{**vars(args), "precision": "fp16x3"}))


@pytest.mark.parametrize("precision", ["bf16x6", "bf16x6_32", "fp32"])
def test_render_call_is_bit_identical_from_launch_to_launch(precision):
    """Four dm_nerf calls on the same 150 rays with other kernels (another arithmetic, the compositing and resampling
    kernels) running in between: every output bit for bit the same.  Guards against reads of memory a kernel does not own --
    the 16x16x32 kernel once read encoding rows of the neighbouring wave into zero-weight k-slots, where NaN bit patterns
    left in LDS by whatever ran before survived as sign-dependent zeros: a few rgb values per launch, on some devices."""
    n = 150
    o, d, _, _ = cases.path_rays(n, seed=41)
    pos, view, coarse, fine, args = nets(13, 0, 1)
    args.precision = precision
    rays = G(np.stack([o, d], 0))
    z = dm.z_val_sample(n, cases.NEAR, cases.FAR, cases.N_SAMPLES)
    other = SimpleNamespace(**{**vars(args), "precision": "fp16x3"})
    with torch.no_grad():
        first = dm.dm_nerf(rays, pos, view, coarse, fine, z, args)
        for _ in range(3):
            dm.dm_nerf(rays[:, :64], pos, view, coarse, fine, z[:64], other)
            again = dm.dm_nerf(rays, pos, view, coarse, fine, z, args)
            for k in first:
                np.testing.assert_array_equal(H(again[k]), H(first[k]), err_msg=k)
    assert all(np.isfinite(H(v)).all() for v in first.values())


@pytest.mark.parametrize("ins_num", [13, 40, 93, 127])
def test_tile_loop_gives_every_tile_the_bits_it_gets_alone(ins_num):
    """field_fwd_x6w_kernel runs one workgroup per CU that walks the launch's 128-point tiles, the weight stream wrapping
    from a tile's last stage into the next tile's first (slot parity carried over tiles when the stage count is odd: 6 and
    8 instance tiles).  A launch in which every workgroup walks 3 - 4 tiles (ragged tail included) against the same rays
    in launches of fewer tiles than the device has CUs (one tile per workgroup, nothing wraps): bit for bit the same
    rows, full and density-only."""
    rng = np.random.default_rng(ins_num)
    model = dm.DM_NeRF(8, 256, 63, 27, [4], ins_num).to(DEV)
    model.load_state_dict(oracle.make_params(ins_num, 5))
    blob = model.blob("bf16x6")
    n, s = 601, 191                                                # 114 791 points = 897 tiles, the last one ragged
    o = G(rng.uniform(-1, 1, size=(n, 3)).astype(np.float32))
    d = G(rng.standard_normal((n, 3)).astype(np.float32))
    z = G(np.sort(rng.uniform(cases.NEAR, cases.FAR, size=(n, s)).astype(np.float32), -1))
    cus = torch.cuda.get_device_properties(DEV).multi_processor_count
    step = max(1, (min(cus, 200) * 128) // s - 1)                  # rays per small launch: fewer tiles than CUs
    assert (n * s + 127) // 128 > 3 * cus or cus > 256
    for sigma_only in (False, True):
        whole = ops.field_rays(o, d, z, blob, ins_num, "bf16x6", sigma_only=sigma_only)
        for lo in range(0, n, step):
            part = ops.field_rays(o[lo:lo + step], d[lo:lo + step], z[lo:lo + step], blob, ins_num, "bf16x6", sigma_only=sigma_only)
            assert torch.equal(part, whole[lo:lo + step]), (sigma_only, lo)
        if sigma_only:
            assert torch.equal(whole[..., 3], full[..., 3])
            assert float(whole[..., :3].abs().max()) == 0.0 and float(whole[..., 4:].abs().max()) == 0.0
        else:
            full = whole
            assert bool(torch.isfinite(full).all())


# ---- the one-call C entry for a chunk ----------------------------------------------------------------------
@pytest.mark.parametrize("precision,stochastic", [("fp32", False), ("bf16x3", True), ("bf16x6", True), ("bf16x6", False)])
def test_render_rays_entry_point_equals_the_chained_calls(precision, stochastic):
    """dmn_render_rays (coarse -> composite -> resample -> fine -> composite in one C call) against dm_nerf(), which
    chains the same kernels from Python: bit-identical, and the workspace holds raw_* / z_vals_fine."""
    from dm_nerf_amd._lib import load, ptr, current_stream, check
    n, ins_num = 150, 13
    o, d, t_rand, u = cases.path_rays(n, seed=41)
    pos, view, coarse, fine, args = nets(ins_num, 0, 1)
    args = SimpleNamespace(**{**vars(args), "precision": precision, "perturb": 1.0 if stochastic else 0.0,
                              "t_rand": G(t_rand), "u": G(u)})
    z = dm.z_val_sample(n, cases.NEAR, cases.FAR, cases.N_SAMPLES)
    with torch.no_grad():
        want = dm.dm_nerf(G(np.stack([o, d], 0)), pos, view, coarse, fine, z, args)
    lib = load()
    S, Ni, C = cases.N_SAMPLES, cases.N_IMPORTANCE, 5 + ins_num
    ws = torch.empty(lib.dmn_render_workspace_floats(n, S, Ni, ins_num), device=DEV)
    out = {k: torch.empty(n, w, device=DEV) for k, w in (("rgb_fine", 3), ("depth_fine", 1), ("ins_fine", ins_num),
                                                         ("rgb_coarse", 3), ("depth_coarse", 1), ("ins_coarse", ins_num))}
    ro, rd = G(o), G(d)
    zrow = z[:1].contiguous()                                   # the lattice as one shared row (stride 0)
    tr, uu = (G(t_rand), G(u)) if stochastic else (None, None)
    check(lib.dmn_render_rays(ptr(ro), ptr(rd), ptr(zrow), 0, ptr(tr), ptr(uu), Ni if stochastic else 0, n, S, Ni,
                              ptr(coarse.blob(precision)), ptr(fine.blob(precision)), ins_num,
                              ops.inference_precision_code(precision, ins_num), ptr(ws), ptr(out["rgb_fine"]), ptr(out["depth_fine"]),
                              ptr(out["ins_fine"]), ptr(out["rgb_coarse"]), ptr(out["depth_coarse"]),
                              ptr(out["ins_coarse"]), ptr(ops.status_word(DEV)), current_stream()), "dmn_render_rays")
    for k, t in out.items():
        np.testing.assert_array_equal(H(t).reshape(H(want[k]).shape), H(want[k]), err_msg=k)
    off = n * S
    np.testing.assert_array_equal(H(ws[off:off + n * S * C]).reshape(n, S, C), H(want["raw_coarse"]))
    off += n * S * C + n * S
    np.testing.assert_array_equal(H(ws[off:off + n * (S + Ni)]).reshape(n, S + Ni), H(want["z_vals_fine"]))
    off += n * (S + Ni)
    np.testing.assert_array_equal(H(ws[off:off + n * (S + Ni) * C]).reshape(n, S + Ni, C), H(want["raw_fine"]))


# ---- error behaviour ------------------------------------------------------------------------------
def test_errors_are_python_exceptions():
    pos, view, coarse, fine, args = nets(13, 0, 1)
    rays = torch.zeros(2, 4, 3, device=DEV)
    z = dm.z_val_sample(4, 4.0, 15.0, 64)
    with pytest.raises(NotImplementedError):                     # module call on encodings is inference-only
        coarse(torch.zeros(4, 90, device=DEV))
    with pytest.raises(NotImplementedError):
        dm.manipulator_render(torch.zeros(4, 64, 18, device=DEV, requires_grad=True), z, rays[1])
    with pytest.raises(ValueError):
        dm.render_train(torch.zeros(4, 64, 18, device=DEV), torch.zeros(4, 32, device=DEV), rays[1])
    with pytest.raises(TypeError):
        dm.render_train(torch.zeros(4, 64, 18, device=DEV, dtype=torch.float64), z, rays[1])
    with pytest.raises(NotImplementedError):
        dm.DM_NeRF(8, 128, 63, 27, [4], 13)
