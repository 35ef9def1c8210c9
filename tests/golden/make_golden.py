#!/usr/bin/env python3
"""Generate the golden vectors under ``tests/golden/*.npz`` by running the REFERENCE itself.

Run only in the build container (needs ``/root/reference``; the reference never travels):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

It imports ``networks.{dm_nerf,helpers,render}`` from the reference tree, feeds them the
seeded inputs of ``cases.py`` and synthetic weights from ``oracle.make_params`` (weights
are regenerated from the seed on the test side, so they are not stored), and writes the
reference's outputs.  Stochastic paths are made reproducible by substituting
``torch.rand`` with a queue of recorded tensors while the reference runs.

``config.create_nerf`` (needs configargparse) and ``tester.render_test`` (needs cv2/lpips)
do not import in this image; their few relevant lines are re-enacted below from
config.py:150-157 and networks/tester.py:57-85.
"""
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True
REF = os.environ.get("DMNERF_REFERENCE", "/root/reference")
sys.path.insert(0, REF)

import cases  # noqa: E402
from oracle import dmnerf_cpu as oracle  # noqa: E402  (only make_params is used)

from networks import dm_nerf as ref_model  # noqa: E402
from networks import helpers as ref_helpers  # noqa: E402
from networks import render as ref_render  # noqa: E402

torch.autograd.set_detect_anomaly(False)   # undo the import side effect of networks/dm_nerf.py:5
torch.manual_seed(0)


class RandQueue:
    """Stand-in for ``torch.rand`` that hands out pre-recorded tensors in call order."""

    def __init__(self, tensors):
        self.tensors = list(tensors)
        self._orig = torch.rand

    def __call__(self, *size, **kw):
        t = self.tensors.pop(0)
        shape = tuple(size[0]) if len(size) == 1 and not isinstance(size[0], int) else tuple(size)
        assert tuple(t.shape) == shape, (t.shape, shape)
        return t

    def __enter__(self):
        torch.rand = self
        return self

    def __exit__(self, *exc):
        torch.rand = self._orig
        assert not self.tensors, "recorded draws left unused"


def build_nets(ins_num, seed_c, seed_f):
    """config.py:150-157 re-enacted."""
    pos_enc, pos_ch = ref_model.get_embedder(10, 0)
    dir_enc, dir_ch = ref_model.get_embedder(4, 0)
    nets = []
    for seed in (seed_c, seed_f):
        net = ref_model.DM_NeRF(8, 256, pos_ch, dir_ch, [4], ins_num)
        net.load_state_dict(oracle.make_params(ins_num, seed))
        net.eval()
        nets.append(net)
    return pos_enc, dir_enc, nets[0], nets[1]


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **{k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v))
                                 for k, v in arrays.items()})
    print(f"{name}: {os.path.getsize(path) / 1024:.0f} KiB")


def g1_rays():
    H, W, Ks, c2w = cases.camera_cases()
    out = {"input_digest": cases.digest(c2w, *[Ks[k] for k in sorted(Ks)])}
    for name, K in Ks.items():
        o, d = ref_helpers.get_rays_k(H, W, K, torch.from_numpy(c2w))
        out[f"{name}_o"], out[f"{name}_d"] = o, d
    # the training call passes poses[i,:3,:4] (train_dmsr.py:27)
    o, d = ref_helpers.get_rays_k(H, W, Ks["dmsr"], torch.from_numpy(c2w[:3, :4]))
    out["dmsr34_o"], out["dmsr34_d"] = o, d
    save("g1_rays.npz", **out)


def g2_encode():
    x, v = cases.encoder_points()
    pos_enc, _ = ref_model.get_embedder(10, 0)
    dir_enc, _ = ref_model.get_embedder(4, 0)
    save("g2_encode.npz", input_digest=cases.digest(x, v),
         pos=pos_enc.embed(torch.from_numpy(x)), dir=dir_enc.embed(torch.from_numpy(v)))


def g3_field():
    x, v = cases.field_rows()
    out = {"input_digest": cases.digest(x, v)}
    for ins_num, seed in ((13, 0), (93, 3), (7, 4)):
        pos_enc, dir_enc, net, _ = build_nets(ins_num, seed, seed)
        emb = torch.cat([pos_enc.embed(torch.from_numpy(x)), dir_enc.embed(torch.from_numpy(v))], -1)
        with torch.no_grad():
            out[f"raw_ins{ins_num}"] = net(emb)
    save("g3_field.npz", **out)


def g4_composite():
    out = {}
    for tag, (n, s, c, seed) in {"a": (32, 64, 18, 141), "b": (16, 192, 18, 142), "c": (8, 64, 98, 143),
                                 "d": (8, 100, 11, 144)}.items():
        raw, z, d = cases.composite_case(n, s, c, seed)
        rgb, w, depth, ins = ref_render.render_train(torch.from_numpy(raw), torch.from_numpy(z), torch.from_numpy(d))
        out.update({f"{tag}_digest": cases.digest(raw, z, d), f"{tag}_rgb": rgb, f"{tag}_w": w,
                    f"{tag}_depth": depth, f"{tag}_ins": ins})
    save("g4_composite.npz", **out)


def g5_sampler():
    bins, w, u = cases.sampler_case()
    det = ref_helpers.sample_pdf(torch.from_numpy(bins), torch.from_numpy(w), cases.N_IMPORTANCE, det=True)
    with RandQueue([torch.from_numpy(u)]):
        rnd = ref_helpers.sample_pdf(torch.from_numpy(bins), torch.from_numpy(w), cases.N_IMPORTANCE, det=False)
    det40 = ref_helpers.sample_pdf(torch.from_numpy(bins), torch.from_numpy(w), 40, det=True)
    save("g5_sampler.npz", input_digest=cases.digest(bins, w, u), det=det, rnd=rnd, det40=det40)


def g6_path():
    n = 128
    o, d, t_rand, u = cases.path_rays(n)
    rays = torch.from_numpy(np.stack([o, d], 0))
    pos_enc, dir_enc, coarse, fine = build_nets(13, 0, 1)
    z = ref_helpers.z_val_sample(n, cases.NEAR, cases.FAR, cases.N_SAMPLES)
    out = {"input_digest": cases.digest(o, d, t_rand, u), "z_lattice": z[0]}
    keep = ("rgb_fine", "ins_fine", "depth_fine", "z_vals_fine", "rgb_coarse", "ins_coarse", "depth_coarse",
            "z_vals_coarse")
    with torch.no_grad():
        args = SimpleNamespace(perturb=0.0, N_importance=cases.N_IMPORTANCE, is_train=False, N_ins=None)
        res = ref_render.dm_nerf(rays, pos_enc, dir_enc, coarse, fine, z, args)
        for k in keep:
            out[f"det_{k}"] = res[k]
        out["det_raw_coarse_head"] = res["raw_coarse"][:8]
        out["det_raw_fine_head"] = res["raw_fine"][:8]
        args = SimpleNamespace(perturb=1.0, N_importance=cases.N_IMPORTANCE, is_train=True, N_ins=None)
        with RandQueue([torch.from_numpy(t_rand), torch.from_numpy(u)]):
            res = ref_render.dm_nerf(rays, pos_enc, dir_enc, coarse, fine, z, args)
        for k in keep:
            out[f"rnd_{k}"] = res[k]
        out["rnd_raw_fine_head"] = res["raw_fine"][:8]
        # N_importance = 0 ("coarse-only" BASELINE config 1): fine net re-evaluates the coarse depths
        args = SimpleNamespace(perturb=0.0, N_importance=0, is_train=False, N_ins=None)
        res = ref_render.dm_nerf(rays[:, :32], pos_enc, dir_enc, coarse, fine, z[:32], args)
        for k in ("rgb_fine", "ins_fine", "depth_fine", "z_vals_fine"):
            out[f"ni0_{k}"] = res[k]
        # ScanNet crop sampler slice (render.py:88-90)
        args = SimpleNamespace(perturb=0.0, N_importance=cases.N_IMPORTANCE, is_train=True, N_ins=10)
        res = ref_render.dm_nerf(rays[:, :32], pos_enc, dir_enc, coarse, fine, z[:32], args)
        out["nins_ins_fine"], out["nins_ins_coarse"] = res["ins_fine"], res["ins_coarse"]
    save("g6_path.npz", **out)


def g8_image():
    """networks/tester.py:57-85 re-enacted at 20x24 with N_test=100 (ragged tail of 80)."""
    H, W, n_test = 20, 24, 100
    K, c2w = cases.study_intrinsics(H, W), cases.study_c2w()
    pos_enc, dir_enc, coarse, fine = build_nets(13, 0, 1)
    args = SimpleNamespace(perturb=0.0, N_importance=cases.N_IMPORTANCE, is_train=False, N_ins=None,
                           N_test=n_test, near=cases.NEAR, far=cases.FAR, N_samples=cases.N_SAMPLES)
    with torch.no_grad():
        rays_o, rays_d = ref_helpers.get_rays_k(H, W, K, torch.Tensor(c2w))
        flat = torch.stack([rays_o.reshape(-1, 3).float(), rays_d.reshape(-1, 3).float()], 0)
        rgb_parts, ins_parts = [], []
        for piece in torch.split(flat, n_test, dim=1):          # 100,100,100,100,80 rays
            z = ref_helpers.z_val_sample(piece.shape[1], args.near, args.far, args.N_samples)
            info = ref_render.dm_nerf(piece, pos_enc, dir_enc, coarse, fine, z, args)
            rgb_parts.append(info["rgb_fine"])
            ins_parts.append(info["ins_fine"])
    save("g8_image.npz", rgb=torch.cat(rgb_parts).reshape(H, W, 3), ins=torch.cat(ins_parts).reshape(H, W, -1))


def g7_gradients():
    """One backward pass of the reference through dm_nerf with the stochastic path (recorded draws):
    gradients of a fixed linear probe of the outputs w.r.t. every parameter of both networks."""
    n = 48
    o, d, t_rand, u = cases.path_rays(n, seed=18)
    rays = torch.from_numpy(np.stack([o, d], 0))
    pos_enc, dir_enc, coarse, fine = build_nets(13, 0, 1)
    coarse.train(); fine.train()
    z = ref_helpers.z_val_sample(n, cases.NEAR, cases.FAR, cases.N_SAMPLES)
    args = SimpleNamespace(perturb=1.0, N_importance=cases.N_IMPORTANCE, is_train=True, N_ins=None)
    with RandQueue([torch.from_numpy(t_rand), torch.from_numpy(u)]):
        res = ref_render.dm_nerf(rays, pos_enc, dir_enc, coarse, fine, z, args)
    probe = cases.probe_weights({k: tuple(res[k].shape) for k in cases.PROBE_KEYS})
    loss = sum((res[k] * torch.from_numpy(probe[k])).sum() for k in cases.PROBE_KEYS)
    loss.backward()
    out = {"input_digest": cases.digest(o, d, t_rand, u), "loss": loss.detach(),
           "z_vals_fine": res["z_vals_fine"].detach(), "z_vals_coarse": res["z_vals_coarse"].detach()}
    for tag, net in (("c", coarse), ("f", fine)):
        norms = []
        for name, p in net.named_parameters():
            g = p.grad.detach().numpy()
            norms.append(np.linalg.norm(g.astype(np.float64)))
            out[f"{tag}.{name}"] = g[cases.grad_digest_slices(name, g.shape)]
        out[f"{tag}.norms"] = np.asarray(norms)
    save("g7_gradients.npz", **out)


def g9_selectors():
    """get_select_full / get_select_crop of the reference with its np.random.choice draws recorded."""
    c = cases.selector_case()
    draws = []

    def fake_choice(a, size=None, replace=True):
        return draws.pop(0)

    orig = np.random.choice
    np.random.choice = fake_choice
    try:
        draws[:] = [c["full_pick"]]
        tc, ti, rays = ref_helpers.get_select_full(torch.from_numpy(c["rgb"]), torch.from_numpy(c["pose"]), c["K"],
                                                   torch.from_numpy(c["label"]), c["n_train"])
        draws[:] = [c["labeled_pick"], c["unlabeled_pick"]]
        ctc, cti, crays, n_ins = ref_helpers.get_select_crop(torch.from_numpy(c["rgb"]), torch.from_numpy(c["pose"]),
                                                             c["K"], torch.from_numpy(c["label"]), c["ins_index"],
                                                             torch.from_numpy(c["crop_mask"]), c["n_train"])
    finally:
        np.random.choice = orig
    save("g9_selectors.npz", input_digest=cases.digest(c["rgb"], c["label"], c["pose"], c["full_pick"], c["ins_index"]),
         full_target_c=tc, full_target_i=ti, full_rays=rays, crop_target_c=ctc, crop_target_i=cti, crop_rays=crays,
         crop_n_ins=n_ins)


def g10_penalizer():
    """ins_penalizer (networks/penalizer.py:58-62) and the gradient autograd gives it w.r.t. raw."""
    from networks import penalizer as ref_penalizer
    out = {}
    for tag, (n, s, c, tol, width, seed) in cases.PENALIZER_CASES.items():
        raw, z, depth, d = cases.penalizer_case(n, s, c, seed)
        raw_t = torch.from_numpy(raw).requires_grad_(True)
        loss = ref_penalizer.ins_penalizer(raw_t, torch.from_numpy(z), torch.from_numpy(depth), torch.from_numpy(d),
                                           SimpleNamespace(tolerance=tol, deta_w=width))
        loss.backward()
        out.update({f"{tag}_digest": cases.digest(raw, z, depth, d), f"{tag}_loss": loss.detach().reshape(()),
                    f"{tag}_grad": raw_t.grad})
    save("g10_penalizer.npz", **out)


def g11_criterion():
    """ins_criterion (networks/evaluator.py:19-38) with its Hungarian matching, and the gradient w.r.t. the map."""
    from networks import evaluator as ref_evaluator
    out = {}
    for tag, (n, ins_num, present, seed) in cases.CRITERION_CASES.items():
        pred, labels = cases.criterion_case(n, ins_num, present, seed)
        pred_t = torch.from_numpy(pred).requires_grad_(True)
        total, ce, inv, siou = ref_evaluator.ins_criterion(pred_t, torch.from_numpy(labels), ins_num)
        total.sum().backward()
        out.update({f"{tag}_digest": cases.digest(pred, labels), f"{tag}_total": total.detach().reshape(()),
                    f"{tag}_ce": ce.detach().reshape(()), f"{tag}_inv": inv.detach().float().reshape(()),
                    f"{tag}_siou": siou.detach().reshape(()), f"{tag}_grad": pred_t.grad})
    save("g11_criterion.npz", **out)


def _reference_manipulator():
    """The four hot-path functions of networks/manipulator.py:18-205, compiled from the reference's own source text.
    The module does not import here (its top-level imports :3-15 need imageio / lpips / cv2 / skimage, absent from the
    image), but these functions only touch ``torch``, ``F`` and the reference's ``sample_pdf`` -- so their definitions
    are taken out of the parsed file and executed in a namespace holding exactly those; nothing is stubbed."""
    import ast
    import torch.nn.functional as F
    path = os.path.join(REF, "networks", "manipulator.py")
    wanted = ("exchanger", "manipulator_render", "manipulator_nerf", "manipulator")
    tree = ast.parse(open(path).read(), filename=path)
    body = [node for node in tree.body if isinstance(node, ast.FunctionDef) and node.name in wanted]
    assert [node.name for node in body] == list(wanted)
    ns = {"torch": torch, "F": F, "sample_pdf": ref_helpers.sample_pdf}
    exec(compile(ast.Module(body=body, type_ignores=[]), path, "exec"), ns)
    return ns


def g12_manipulator():
    """exchanger / manipulator_render / manipulator_nerf / manipulator of the reference (row N3)."""
    ns = _reference_manipulator()
    out = {}
    T = torch.from_numpy
    # ---- exchanger: label surgery, edited in place
    for tag, (n, s, c, moves, seed) in cases.EXCHANGE_CASES.items():
        ori_raw, ori_pred, tar_raws, tar_preds = cases.exchange_case(n, s, c, moves, seed)
        got_raw, _, ori_label, tar_label = ns["exchanger"](T(ori_raw.copy()), [T(t) for t in tar_raws], T(ori_pred),
                                                           [T(t) for t in tar_preds], list(moves))
        got = got_raw.numpy()
        # where each output row came from: -1 zeroed, 0 kept, 1 + k taken from target k (first match; small, and with
        # the digest of the full tensor it pins the result bit for bit)
        src = np.zeros((n, s), np.int8)
        for k, t in enumerate(tar_raws):
            src[(src == 0) & (got == t).all(-1) & ~(got == ori_raw).all(-1)] = 1 + k
        src[(got == 0).all(-1) & ~(ori_raw == 0).all(-1)] = -1
        out.update({f"ex_{tag}_digest": cases.digest(ori_raw, ori_pred, *tar_raws, *tar_preds),
                    f"ex_{tag}_out_digest": cases.digest(got), f"ex_{tag}_src": src,
                    f"ex_{tag}_ori_label": ori_label.numpy().astype(np.int16),
                    f"ex_{tag}_tar_label": tar_label.numpy().astype(np.int16)})
    # ---- manipulator_render: compositing that keeps the air channel
    for tag, (n, s, c, seed) in {"a": (32, 64, 18, 141), "b": (16, 192, 18, 142), "c": (8, 64, 98, 143),
                                 "d": (8, 100, 11, 144)}.items():
        raw, z, d = cases.composite_case(n, s, c, seed)
        rgb, w, depth, ins = ns["manipulator_render"](T(raw), T(z), T(d))
        out.update({f"mr_{tag}_rgb": rgb, f"mr_{tag}_w": w, f"mr_{tag}_depth": depth, f"mr_{tag}_ins": ins})
    # ---- manipulator_nerf: lattice and raw rows; manipulator: the two-pass re-render with recorded draws
    o, d, tars, draws = cases.manipulator_case()
    pos_enc, _ = ref_model.get_embedder(10, 0)
    dir_enc, _ = ref_model.get_embedder(4, 0)
    nets = []
    for seed, jitter in cases.MANIPULATOR_NETS:
        net = ref_model.DM_NeRF(8, 256, 63, 27, [4], 13)
        net.load_state_dict(cases.manipulator_params(13, seed, jitter))
        nets.append(net.eval())
    rays = T(np.stack([o, d], 0))
    f_tar = [T(np.stack([a, b], 0)) for a, b in tars]
    margs = SimpleNamespace(N_samples=cases.N_SAMPLES, N_importance=cases.N_IMPORTANCE, near=cases.NEAR, far=cases.FAR,
                            target_label=list(cases.MANIPULATOR_LABELS))
    with torch.no_grad():
        raw, z = ns["manipulator_nerf"](rays[:, :6], pos_enc, dir_enc, nets[0], cases.N_SAMPLES, cases.NEAR, cases.FAR)
        out.update({"mn_z_lattice": z[0], "mn_raw": raw})
        raw2, z2 = ns["manipulator_nerf"](rays[:, :6], pos_enc, dir_enc, nets[1], z_vals=z[:, ::3] * 1.01)
        out["mn_raw_given_z"] = raw2
        trace = {"labels": [], "z": []}
        inner_ex, inner_nerf = ns["exchanger"], ns["manipulator_nerf"]

        def ex_spy(*a):
            res = inner_ex(*a)
            trace["labels"].append((res[2].numpy().astype(np.int16), res[3].numpy().astype(np.int16)))
            return res

        def nerf_spy(*a, **kw):
            res = inner_nerf(*a, **kw)
            trace["z"].append(res[1])
            return res

        ns["exchanger"], ns["manipulator_nerf"] = ex_spy, nerf_spy
        with RandQueue([T(u) for u in draws]):
            final_rgb, final_ins, tar_rgb, tar_acc = ns["manipulator"](pos_enc, dir_enc, nets[0], nets[1], rays, f_tar,
                                                                      margs)
        ns["exchanger"], ns["manipulator_nerf"] = inner_ex, inner_nerf
    out.update({"mp_digest": cases.digest(o, d, *[x for t in tars for x in t], *draws),
                "mp_final_rgb": final_rgb, "mp_final_ins": final_ins, "mp_tar_rgb": tar_rgb, "mp_tar_ins_accum": tar_acc,
                "mp_ex1_ori_label": trace["labels"][0][0], "mp_ex1_tar_label": trace["labels"][0][1],
                "mp_ex2_ori_label": trace["labels"][1][0], "mp_ex2_tar_label": trace["labels"][1][1],
                "mp_z_full_first": trace["z"][1],        # original rays: 64 + 128 depths of pass 1
                "mp_z_final": trace["z"][-2]})           # original rays: 64 + 128 + 128 * objects depths of pass 2
    save("g12_manipulator.npz", **out)


def _reference_maps(rays, dtype):
    """One deterministic dm_nerf call of the reference (networks/render.py:31-96) in `dtype`: the modules, the rays and
    every tensor the reference creates on the way (lattice, linspace draws, the 1e10 sentinel) are of that type; the
    parameters are the same fp32 values (load_state_dict copies them into the modules exactly)."""
    keep = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    try:
        pos_enc, dir_enc, coarse, fine = build_nets(13, 0, 1)
        assert next(coarse.parameters()).dtype == dtype
        n = rays.shape[1]
        z = ref_helpers.z_val_sample(n, cases.NEAR, cases.FAR, cases.N_SAMPLES).to(dtype)
        args = SimpleNamespace(perturb=0.0, N_importance=cases.N_IMPORTANCE, is_train=False, N_ins=None)
        with torch.no_grad():
            res = ref_render.dm_nerf(rays.to(dtype), pos_enc, dir_enc, coarse, fine, z, args)
        assert res["rgb_fine"].dtype == dtype and res["z_vals_fine"].dtype == dtype, (res["rgb_fine"].dtype, dtype)
        return {k: res[k] for k in ("rgb_fine", "depth_fine", "ins_fine", "z_vals_fine", "rgb_coarse", "depth_coarse")}
    finally:
        torch.set_default_dtype(keep)


def g13_flip():
    """The reference against ITSELF behind the sampler (networks/helpers.py:139-153, networks/render.py:66-70): the same
    deterministic dm_nerf call in fp32 and in fp64 (fp64 modules holding the same fp32 parameters) on the ray sets the
    end-to-end gates use -- the 2048 random pixels of the 800x800 study camera (tests/test_gpu_parity.py,
    test_flip_rate_*) and the 1024 rays of smoke()'s 64x64 camera.  Stored: the rays (the reference's get_rays_k on the
    CPU, so that both sides start from the same bits), the fp32 maps and merged depths, the fp64 maps, and the
    per-ray deviation of the reference's fp32 result from its fp64 result."""
    out = {}
    for tag, (hw, pick) in cases.flip_ray_sets().items():
        K, c2w = cases.study_intrinsics(hw, hw), cases.study_c2w()
        ro, rd = ref_helpers.get_rays_k(hw, hw, K, torch.Tensor(c2w))
        rays = torch.stack([ro.reshape(-1, 3).float()[pick], rd.reshape(-1, 3).float()[pick]], 0)
        r32 = _reference_maps(rays, torch.float32)
        r64 = _reference_maps(rays, torch.float64)
        out[f"{tag}_rays"] = rays
        out[f"{tag}_pick"] = np.asarray(pick, np.int32)
        for k in ("rgb_fine", "depth_fine", "ins_fine", "z_vals_fine", "rgb_coarse", "depth_coarse"):
            out[f"{tag}_{k}"] = r32[k]
        for k in ("rgb_fine", "depth_fine", "ins_fine"):
            out[f"{tag}_f64_{k}"] = r64[k].float()
            a, b = r32[k].double().reshape(rays.shape[1], -1), r64[k].reshape(rays.shape[1], -1)
            per_ray = (a - b).abs().max(-1).values / b.abs().max()
            out[f"{tag}_self_dev_{k}"] = per_ray.float()
            print(f"  {tag} {k}: reference fp32 vs fp64: {float((per_ray > 1e-4).double().mean()):.4f} of {rays.shape[1]} rays "
                  f"beyond 1e-4, worst {float(per_ray.max()):.2e}")
        dz = (r32["z_vals_fine"].double() - r64["z_vals_fine"]).abs().max(-1).values / r64["z_vals_fine"].abs().max()
        out[f"{tag}_self_dev_z_vals_fine"] = dz.float()
    save("g13_flip.npz", **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g2", "g3", "g4", "g5", "g6", "g7", "g8", "g9", "g10", "g11", "g12", "g13"]
    table = {"g1": g1_rays, "g2": g2_encode, "g3": g3_field, "g4": g4_composite, "g5": g5_sampler,
             "g6": g6_path, "g7": g7_gradients, "g8": g8_image, "g9": g9_selectors, "g10": g10_penalizer,
             "g11": g11_criterion, "g12": g12_manipulator, "g13": g13_flip}
    for key in which:
        table[key]()
