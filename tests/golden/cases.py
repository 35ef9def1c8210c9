"""Seeded input builders shared by ``make_golden.py`` (which feeds them to the reference)
and by the tests (which feed them to the oracle and to the HIP path).  Inputs are regenerated
from numpy PCG64 seeds on both sides; the fixtures store an ``input_digest`` so a drift in
the generator would be detected instead of silently comparing different inputs."""
from __future__ import annotations

import hashlib
import math

import numpy as np

NEAR, FAR = 4.0, 15.0          # configs/test/dmsr/study.txt:13-14
N_SAMPLES, N_IMPORTANCE = 64, 128


def digest(*arrays) -> str:
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()[:16]


# ---- G1: cameras -------------------------------------------------------------------
def camera_cases():
    """Three intrinsics conventions at H=6, W=8 plus one non-trivial pose."""
    H, W = 6, 8
    f = 0.5 * W / math.tan(0.5 * 0.6911)
    k_dmsr = np.array([[f, 0, W * 0.5], [0, -f, H * 0.5], [0, 0, -1]], dtype=np.float64)
    k_replica = np.array([[W / 2.0, 0, (W - 1) / 2.0], [0, W / 2.0, (H - 1) / 2.0], [0, 0, 1]], dtype=np.float64)
    k_scannet = np.array([[577.87, 0, 319.5, 0], [0, 577.87, 239.5, 0], [0, 0, 1, 0], [0, 0, 0, 1]], dtype=np.float64)
    rng = np.random.default_rng(11)
    q, _ = np.linalg.qr(rng.standard_normal((3, 3)))
    c2w = np.eye(4, dtype=np.float32)
    c2w[:3, :3] = q.astype(np.float32)
    c2w[:3, 3] = (0.3, -0.2, 6.0)
    return H, W, {"dmsr": k_dmsr, "replica": k_replica, "scannet": k_scannet}, c2w


# ---- G2: encoder inputs ------------------------------------------------------------
def encoder_points():
    rng = np.random.default_rng(12)
    x = rng.uniform(-4.0, 4.0, size=(256, 3)).astype(np.float32)
    x[:32] *= 3.75                      # |x| up to 15 -> arguments up to ~7.7e3 rad at 2^9
    x[32:40] = 0.0
    x[40, :] = (15.0, -15.0, 14.999)
    x[41, :] = (1e-6, -1e-6, 3.1415927)
    v = rng.standard_normal((256, 3)).astype(np.float32)
    v /= np.linalg.norm(v, axis=-1, keepdims=True)
    return x, v.astype(np.float32)


# ---- G3: field inputs --------------------------------------------------------------
def field_rows():
    """256 embedded rows are built from these points/dirs by the side under test."""
    rng = np.random.default_rng(13)
    x = rng.uniform(-9.0, 9.0, size=(256, 3)).astype(np.float32)
    v = rng.standard_normal((256, 3)).astype(np.float32)
    v /= np.linalg.norm(v, axis=-1, keepdims=True)
    return x, v.astype(np.float32)


# ---- G4: compositing inputs --------------------------------------------------------
def composite_case(n_rays: int, n_samples: int, channels: int, seed: int):
    rng = np.random.default_rng(seed)
    raw = rng.standard_normal((n_rays, n_samples, channels)).astype(np.float32) * 2.0
    raw[..., 3] *= 4.0                                    # densities: mixture of <0 and large
    raw[0, :, 3] = -1.0                                   # fully transparent ray
    raw[1, :, 3] = 1e4                                    # saturated alpha from the first sample
    raw[2, : n_samples // 2, 3] = -3.0                    # empty then dense
    z = np.sort(rng.uniform(NEAR, FAR, size=(n_rays, n_samples)).astype(np.float32), axis=-1)
    z[3] = np.linspace(NEAR, FAR, n_samples, dtype=np.float32)
    z[4, 10:14] = z[4, 10]                                # repeated depths -> zero-length intervals
    d = rng.standard_normal((n_rays, 3)).astype(np.float32)
    d[5] *= 1e-3
    return raw, z, d


# ---- G10: emptiness-penalizer inputs (raw, z, depth, rays_d) ---------------------------
# tag -> (rays, samples, channels, tolerance, deta_w, seed); 0.05 / 0.05 are the reference configs' values
PENALIZER_CASES = {"a": (24, 64, 18, 0.05, 0.05, 301), "b": (12, 192, 18, 0.05, 0.05, 302),
                   "c": (6, 64, 98, 0.3, 0.2, 303), "d": (7, 100, 11, 1.5, 0.5, 304)}


def penalizer_case(n_rays: int, n_samples: int, channels: int, seed: int):
    rng = np.random.default_rng(seed)
    raw = rng.standard_normal((n_rays, n_samples, channels)).astype(np.float32) * 3.0
    raw[0, :, 4:] = 40.0                                  # saturated logits: log(1 - p + 1e-8) clamps
    raw[1, :, 4:] = -40.0
    z = np.sort(rng.uniform(NEAR, FAR, size=(n_rays, n_samples)).astype(np.float32), axis=-1)
    depth = rng.uniform(NEAR + 1.0, FAR - 1.0, size=n_rays).astype(np.float32)
    depth[2] = NEAR - 0.5                                 # surface in front of every sample: nothing "before"
    depth[3] = FAR + 0.5                                  # behind every sample: everything "before"
    depth[4] = z[4, n_samples // 2]                       # exactly on a sample
    d = rng.standard_normal((n_rays, 3)).astype(np.float32)
    d[5] *= 1e-2
    return raw, z, depth, d


# ---- G11: instance-matching loss inputs (rendered instance map, integer labels) --------
# tag -> (rays, ins_num, labels present, seed)
CRITERION_CASES = {"a": (96, 13, (0, 2, 3, 7, 12), 311), "b": (64, 13, tuple(range(13)), 312),
                   "c": (80, 93, (1, 5, 40, 41, 90, 92), 313), "d": (40, 7, (4,), 314)}


def criterion_case(n_rays: int, ins_num: int, present, seed: int):
    rng = np.random.default_rng(seed)
    labels = rng.choice(np.asarray(present), size=n_rays).astype(np.int64)
    labels[: len(present)] = present                      # every listed label occurs
    pred = rng.uniform(0.02, 0.98, size=(n_rays, ins_num)).astype(np.float32)
    perm = rng.permutation(ins_num)                       # a (shuffled) column responds to each class
    for k, lab in enumerate(present):
        hit = labels == lab
        pred[hit, perm[k]] = rng.uniform(0.6, 0.999, size=int(hit.sum())).astype(np.float32)
    pred[0, :] = 1.0                                      # saturated predictions: the 1e-8 clamps
    pred[1, :] = 0.0
    return pred, labels


# ---- G5: sampler inputs ------------------------------------------------------------
def sampler_case(n_rays: int = 48, seed: int = 15):
    rng = np.random.default_rng(seed)
    z = np.linspace(NEAR, FAR, N_SAMPLES, dtype=np.float32)[None].repeat(n_rays, 0)
    z[n_rays // 2:] += rng.uniform(0, 0.05, size=(n_rays - n_rays // 2, N_SAMPLES)).astype(np.float32)
    z = np.sort(z, -1)
    bins = 0.5 * (z[:, 1:] + z[:, :-1])
    w = rng.uniform(0, 1, size=(n_rays, N_SAMPLES - 2)).astype(np.float32) ** 8
    w[0] = 0.0                                            # all-zero weights -> uniform pdf
    w[1] = 0.0; w[1, 20] = 1.0                            # single peak
    w[2] = 0.0; w[2, 0] = 0.7; w[2, -1] = 0.3             # mass at both ends
    w[3] = 1.0                                            # flat
    u = rng.uniform(0, 1, size=(n_rays, N_IMPORTANCE)).astype(np.float32)
    u[4, :4] = (0.0, 0.99999994, 0.5, 1e-7)
    return bins.astype(np.float32), w, u


# ---- G6/G8: rays for whole-path cases ----------------------------------------------
def study_intrinsics(H: int, W: int) -> np.ndarray:
    f = 0.5 * W / math.tan(0.5 * 0.6911)
    return np.array([[f, 0.0, W / 2.0], [0.0, -f, H / 2.0], [0.0, 0.0, -1.0]], dtype=np.float64)


def study_c2w() -> np.ndarray:
    c2w = np.eye(4, dtype=np.float32)
    c2w[:3, 3] = (0.3, -0.2, 6.0)
    return c2w


def flip_ray_sets():
    """G13: the ray sets of the end-to-end gates, as (image side, flat pixel indices) of the square study camera:
    the 2048 random pixels of the 800x800 image (tests/test_gpu_parity.py, flip-rate tests) and the 1024 pixels of the
    64x64 image smoke() renders (every fourth)."""
    return {"t2048": (800, np.sort(np.random.default_rng(31).choice(800 * 800, size=2048, replace=False))),
            "s1024": (64, np.arange(0, 64 * 64, 4)[:1024])}


def path_rays(n_rays: int, seed: int = 16):
    """Random subset of the 800x800 'study' camera's rays (computed in float64 -> fp32 here so
    that ray generation is not part of this case)."""
    H = W = 800
    K = study_intrinsics(H, W)
    rng = np.random.default_rng(seed)
    pix = rng.choice(H * W, size=n_rays, replace=False)
    r, c = pix // W, pix % W
    d = np.stack([(c - K[0, 2]) / K[0, 0], (r - K[1, 2]) / K[1, 1], np.full(n_rays, K[2, 2])], -1)
    o = np.broadcast_to(study_c2w()[:3, 3], d.shape)
    t_rand = rng.uniform(0, 1, size=(n_rays, N_SAMPLES)).astype(np.float32)
    u = rng.uniform(0, 1, size=(n_rays, N_IMPORTANCE)).astype(np.float32)
    return o.astype(np.float32).copy(), d.astype(np.float32), t_rand, u


# ---- G7: gradient probe --------------------------------------------------------------
PROBE_KEYS = ("rgb_fine", "ins_fine", "depth_fine", "rgb_coarse", "ins_coarse", "depth_coarse", "raw_fine",
              "raw_coarse")


def probe_weights(shapes: dict, seed: int = 17) -> dict:
    """Fixed random linear functional of the render outputs: loss = sum_k <out_k, P_k>.  It reaches every
    gradient path the training losses use (colour and instance maps, depth, and raw through the penalizer)
    without dragging the loss heads (out of scope) into the fixture."""
    rng = np.random.default_rng(seed)
    return {k: rng.standard_normal(shapes[k]).astype(np.float32) for k in PROBE_KEYS}


def grad_digest_slices(name: str, shape) -> tuple:
    """Which part of a gradient tensor the fixture stores: everything for small tensors, a strided
    subsample for the 256x256-class matrices (the per-tensor norms cover the rest)."""
    n = int(np.prod(shape))
    if n <= 4096:
        return (slice(None),) * len(shape)
    return (slice(None, None, 7), slice(None, None, 5))


# ---- G9: training ray selectors -----------------------------------------------------------
def selector_case():
    """A 12x16 image with labels, a pose, a crop rectangle with labelled pixels, and the recorded draws."""
    H, W = 12, 16
    rng = np.random.default_rng(19)
    rgb = rng.uniform(0, 1, size=(H, W, 3)).astype(np.float32)
    label = rng.integers(0, 5, size=(H, W)).astype(np.int64)
    q, _ = np.linalg.qr(rng.standard_normal((3, 3)))
    pose = np.concatenate([q.astype(np.float32), np.array([[0.3], [-0.2], [6.0]], np.float32)], 1)   # 3x4
    K = study_intrinsics(H, W)
    n_train = 40
    full_pick = rng.choice(H * W, size=n_train, replace=False)
    crop_mask = np.zeros((H, W), np.int64)
    crop_mask[2:10, 3:13] = 1
    ins_index = np.nonzero((crop_mask.reshape(-1) == 1) & (rng.uniform(size=H * W) < 0.25))[0]      # labelled pixels
    n_ins = min(int(n_train * 0.3), len(ins_index))
    labeled_pick = rng.choice(len(ins_index), size=n_ins, replace=False)
    n_unlabeled = int(crop_mask.sum()) - n_ins
    unlabeled_pick = rng.choice(n_unlabeled, size=n_train - n_ins, replace=False)
    return dict(H=H, W=W, rgb=rgb, label=label, pose=pose, K=K, n_train=n_train, full_pick=full_pick,
                crop_mask=crop_mask, ins_index=ins_index, labeled_pick=labeled_pick, unlabeled_pick=unlabeled_pick)


# ---- G12: object manipulator (networks/manipulator.py:18-205) -------------------------------
# tag -> (rays, samples, channels, moved labels, seed)
EXCHANGE_CASES = {"a": (48, 64, 18, (3, 7), 55), "b": (12, 192, 98, (40,), 56), "c": (20, 100, 11, (2, 5, 0), 57)}


def exchange_case(n: int, s: int, c: int, moves, seed: int):
    """Per-point raw rows and per-ray composited logits of an original scene and of one transformed scene per moved
    label, biased so that every branch of ``exchanger`` occurs (occluders, fillings, take, wipe, keep), with
    saturated logits on a few points so that ``argmax(sigmoid(.))`` ties resolve to the first channel."""
    rng = np.random.default_rng(seed)

    def scene():
        raw = (rng.standard_normal((n, s, c)) * 3).astype(np.float32)
        for lab in moves:
            raw[rng.random((n, s)) < 0.2, 4 + lab] += 9.0
        raw[0, :8, 4:] = 40.0                                   # sigmoid saturates to 1.0 on every channel
        pred = (rng.standard_normal((n, c - 4)) * 3).astype(np.float32)
        for lab in moves:
            pred[rng.random(n) < 0.3, lab] += 9.0
        return raw, pred

    ori_raw, ori_pred = scene()
    tars = [scene() for _ in moves]
    return ori_raw, ori_pred, [t[0] for t in tars], [t[1] for t in tars]


def manipulator_params(ins_num: int, seed: int, jitter_seed=None):
    """Seeded weights with a softer density and a louder instance head than ``make_params`` gives: rays then end at
    depths 6.5 .. 8.5 with five different point labels along them, which the two-pass re-render needs to do anything.
    ``jitter_seed``: add 2 % noise, so that a "fine" network agrees with the "coarse" one about the scene (as a
    trained pair does) without being the same function."""
    from dm_nerf_amd.synthetic import make_params
    p = make_params(ins_num, seed, sharpen=False)
    p["density_linear.weight"] = p["density_linear.weight"] * 12.0
    p["ins_linear.weight"] = p["ins_linear.weight"] * 10.0
    p["ins_feature_linear.weight"] = p["ins_feature_linear.weight"] * 3.0
    if jitter_seed is not None:
        import torch
        rng = np.random.default_rng(jitter_seed)
        for k in p:
            noise = rng.standard_normal(tuple(p[k].shape)).astype(np.float32)
            p[k] = p[k] + 0.02 * float(p[k].abs().max()) * torch.from_numpy(noise)
    return p


MANIPULATOR_NETS = ((0, None), (0, 5))      # (seed, jitter_seed) of the coarse and the fine network
MANIPULATOR_LABELS = (1, 4)


def manipulator_case(n_rays: int = 24, seed: int = 61):
    """Original rays, one inversely transformed ray set per moved object (a translation; a translation plus a
    rotation about the view axis), and the uniform draws of the 2 + len(labels) ``sample_pdf`` calls."""
    o, d, _, _ = path_rays(n_rays, seed=seed)
    rng = np.random.default_rng(seed + 1)
    ca, sa = math.cos(0.1), math.sin(0.1)
    rot = np.array([[ca, -sa, 0.0], [sa, ca, 0.0], [0.0, 0.0, 1.0]], dtype=np.float32)
    tars = [(o + np.array([0.4, 0.0, 0.0], np.float32), d.copy()),
            ((o + np.array([-0.3, 0.2, 0.0], np.float32)) @ rot.T, d @ rot.T)]
    tars = [(a.astype(np.float32), b.astype(np.float32)) for a, b in tars]
    draws = [rng.uniform(0, 1, size=(n_rays, N_IMPORTANCE)).astype(np.float32) for _ in range(2 + len(tars))]
    return o, d, tars, draws


# ---- BASELINE config 5: ScanNet full resolution --------------------------------------------
SCANNET_H, SCANNET_W = 968, 1296                       # colour frames of a ScanNet scene (loader_scannet.py:98-99)
SCANNET_CROP = (640, 480)                              # crop_width, crop_height (configs/test/scannet/scene0010_00.txt:13-14)
SCANNET_NEAR, SCANNET_FAR = 0.0, 9.5                   # :17-18


def scannet_intrinsics() -> np.ndarray:
    """4x4 ``intrinsic_color.txt``-shaped matrix (values of a typical ScanNet colour camera)."""
    return np.array([[1170.19, 0.0, 647.75, 0.0], [0.0, 1170.19, 483.75, 0.0], [0.0, 0.0, 1.0, 0.0],
                     [0.0, 0.0, 0.0, 1.0]], dtype=np.float64)


def scannet_crop_mask(H: int, W: int, crop_size) -> np.ndarray:
    """The centre rectangle the ScanNet loader marks (datasets/loader_scannet.py:17-24 restated)."""
    mask = np.zeros((H, W), dtype=np.int8)
    new_w, new_h = crop_size
    mh, mw = (H - new_h) // 2, (W - new_w) // 2
    mask[mh:H - mh, mw:W - mw] = 1
    return mask


def scannet_pose(index: int) -> np.ndarray:
    """Camera inside a room-sized volume, looking along +z with a small per-pose turn (OpenCV convention, K22 = 1)."""
    a = 0.15 * index
    c2w = np.eye(4, dtype=np.float32)
    c2w[:3, :3] = np.array([[math.cos(a), 0, math.sin(a)], [0, 1, 0], [-math.sin(a), 0, math.cos(a)]], np.float32)
    c2w[:3, 3] = (0.2 * index - 0.5, 0.1, -1.0)
    return c2w
