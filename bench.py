#!/usr/bin/env python3
"""Headline benchmark: rays/s for an 800x800 render of the DM-SR 'study' configuration
(64 coarse + 128 importance samples, 8x256 field with ins_num = 13 instance head) through the
HIP hot path, plus the roofline of the dominant kernel and the CPU oracle timed beside it.

    python bench.py --gpus N --steps K --warmup W

One step = one full 800x800 image (640,000 rays) rendered through ``dm_nerf_amd.render_image``
(ray generation -> per chunk: coarse field, composite, resample+merge, fine field, composite).
``value`` is measured in fp32-grade arithmetic on the bf16 matrix pipe (``--precision bf16x6``: both operands of every
fp32 product split EXACTLY into three bf16 values, six MFMA products, fp32 accumulation -- no operand bit is dropped;
``arithmetic_check`` in the output line measures it, the exact-fp32 MFMA kernel and the reference's own fp32 result
against an fp64 evaluation).  The exact-fp32 MFMA mode (``exact_fp32_mode``) and the two-way split modes (three 16-bit
MFMAs per product, ``split_precision_modes``) are timed over the same --steps right after it and reported beside it,
all with their deviation statistics against the CPU oracle.
Inputs are synthetic (no dataset / checkpoint exists offline): DM-SR style intrinsics, orbit
poses, PCG64-seeded weights.  With N > 1 -- one process per GPU, started by
torch.distributed.run or, when no launcher set WORLD_SIZE, by this script itself as a child process -- every rank
renders its own pose (``--scaling weak``, the default: no data-path collective) or a contiguous ray range of ONE image
(``--scaling strong``: ray-sharded, the finished maps all-gathered inside the timed region and checked bit for bit
against the one-rank image; networks/tester.py:54-85 is the loop that is cut).

Prints ONE JSON line on rank 0."""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def _self_launch_if_needed():
    """`python bench.py --gpus N` without a launcher: start N ranks under torch.distributed.run as a child process and
    exit with its status.  Runs before torch is imported -- the parent never touches the GPU (dm_nerf_amd/launch.py is
    loaded by path: importing the package would import torch)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("_dmn_launch", os.path.join(ROOT, "dm_nerf_amd", "launch.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    if mod.needs_self_launch(sys.argv[1:]):
        sys.exit(mod.self_launch(os.path.abspath(__file__), sys.argv[1:]))


if __name__ == "__main__":
    _self_launch_if_needed()

import numpy as np  # noqa: E402
import torch  # noqa: E402

import dm_nerf_amd as dm  # noqa: E402
from dm_nerf_amd import ops  # noqa: E402

H = W = 800
INS_NUM = 13
N_SAMPLES, N_IMPORTANCE = 64, 128
MACS_PER_POINT = 693_504                       # BASELINE.md section 3 (networks/dm_nerf.py:65-78)
FLOP_PER_POINT = 2 * MACS_PER_POINT
PEAK_F32_MFMA_TFLOPS = 157.3                   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_16BIT_MFMA_TFLOPS = 2500.0                # MI355X_MICROARCH.md: bf16 / f16 MFMA, dense (no sparsity)
# measured with tools/mfma_peak.hip (register-only MFMA loop on all 256 CUs, random operands): the board power limit
# pulls the engine clock to 1.70 / 1.87 GHz for f16 / bf16; fp32 MFMA holds 2.39 GHz
SUSTAINED_MFMA_TFLOPS = {"fp16x3": 1738.0, "bf16x3": 1915.0, "bf16x6": 1915.0, "fp32": 154.6}
MFMA_PRODUCTS = {"fp32": 1, "fp16x3": 3, "bf16x3": 3, "bf16x6": 6}      # matrix-pipe products issued per algorithmic product
PEAK_HBM_GBS = 8000.0                          # MI355X_MICROARCH.md: HBM3E spec
CPU_CHUNK = 4096                               # the reference's N_test (configs/test/dmsr/study.txt:10)


def composite_bytes(n_rays: int, s: int, c: int) -> int:
    # raw S*C*4 + z S*4 + d 12 in; w S*4 + rgb 12 + depth 4 + ins 4*ins_num out
    return n_rays * (s * c * 4 + s * 4 + 12 + s * 4 + 12 + 4 + 4 * (c - 5))


def host_cores() -> int:
    """CPU threads for the baseline leg: the process's share of the host (cgroup quota or affinity),
    capped at 16 -- the CPU share of a 1-GPU box; os.cpu_count() reports all 256 host threads there
    and oversubscribing them made the oracle 8x slower.  DMNERF_CPU_THREADS overrides."""
    if os.environ.get("DMNERF_CPU_THREADS"):
        return max(1, int(os.environ["DMNERF_CPU_THREADS"]))
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 16))


def synthetic_scene(device, near=4.0, far=15.0):
    """Weights and cameras of the synthetic 'study' workload.  The weight tensors come from the
    seeded generator in dm_nerf_amd.synthetic (data only, shared with the parity tests)."""
    from dm_nerf_amd import synthetic as oracle
    args = dm.default_args(ins_num=INS_NUM, N_samples=N_SAMPLES, N_importance=N_IMPORTANCE, N_test=4096,
                           near=near, far=far, device=device)
    pos, view, coarse, fine, args = dm.create_nerf(args)
    params = oracle.make_params(INS_NUM, 0), oracle.make_params(INS_NUM, 1)
    coarse.load_state_dict(params[0])
    fine.load_state_dict(params[1])
    K = oracle.dmsr_intrinsics(H, W)
    poses = [oracle.study_pose(i) for i in range(8)]
    return (pos, view, coarse.eval(), fine.eval(), args), K, poses, params


def training_leg(nerf, K, pose, device, n_rays: int = 4096, steps: int = 4):
    """BASELINE config 4 on one GPU, outside the timed region: `steps` optimisation steps of `n_rays` random rays of
    the same camera (stochastic path, colour + instance-matching + emptiness losses, Adam), hand-written forward and
    backward kernels: forward AND backward in bf16x6 (fp32-grade products on the bf16 pipe both ways: the config-4
    number), in exact fp32 MFMA, and -- the explicit opt-in -- in bf16x3 (16 operand bits)."""
    from dm_nerf_amd import trainer
    pos, view, coarse, fine, args = nerf
    state = [{k: v.clone() for k, v in m.state_dict().items()} for m in (coarse, fine)]
    targs = dm.default_args(**{**vars(args), "perturb": 1.0, "is_train": True, "tolerance": 0.05, "deta_w": 0.05})
    g = torch.Generator(device=device).manual_seed(2)
    pick = torch.randperm(H * W, device=device, generator=g)[:n_rays]
    ro, rd = dm.get_rays_k(H, W, K, pose)
    rays = torch.stack([ro.reshape(-1, 3)[pick], rd.reshape(-1, 3)[pick]], 0)
    target = torch.rand(n_rays, 3, device=device, generator=g)
    labels = torch.randint(0, INS_NUM, (n_rays,), device=device, generator=g)
    z = dm.z_val_sample(n_rays, targs.near, targs.far, targs.N_samples)
    result = {"rays_per_step": n_rays, "losses": "mse + ins_criterion + ins_penalizer (coarse and fine), Adam 5e-4; the "
              "Hungarian step of ins_criterion runs as a kernel (assign='device': scipy's algorithm in fp64, bit-identical "
              "matching, tests/test_gpu_parity.py) so that the step never waits for the device",
              "backward": "HIP: backward-data + weight-gradient kernels in the arithmetic of the forward's grade "
                          "(ops.backward_precision: bf16x6 -> bf16x6, fp32 -> exact fp32 MFMA, bf16x3 -> bf16x3), compositing backward"}
    coarse.train(); fine.train()
    for prec in ("bf16x6", "fp32", "bf16x3"):
        targs.precision = prec
        opt = trainer.make_optimizer(coarse, fine)
        # losses are read back on the last timed step only (the reference reads them every i_print steps, train_dmsr.py:75)
        run = lambda i, report=False: trainer.train_step(
            i, rays, target, (pos, view, coarse, fine, targs), z, opt,
            ins_loss=trainer.instance_loss(labels, INS_NUM, assign="device"),
            penalizer=lambda raw, zz, depth: dm.ins_penalizer(raw, zz, depth, rays[1], targs), report=report)
        run(0)
        torch.cuda.synchronize()
        ops.PROFILE = []
        t0 = time.perf_counter()
        for i in range(steps):
            last = run(i + 1, report=i == steps - 1)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        prof, ops.PROFILE = ops.PROFILE, None
        entry = {"ms_per_step": dt * 1e3, "rays_per_s": n_rays / dt, "loss_total": last["total"],
                 "backward_precision": ops.backward_precision(prec), "kernels_ms_per_step": {}}
        for name in ("field_fwd_save", "field_bwd_data", "field_bwd_weights"):
            _, secs, _ = kernel_stats(prof, name)
            entry["kernels_ms_per_step"][name] = secs / steps * 1e3
        # the HBM-bound part of the step: the weight-gradient GEMMs read both records (save: activations, gradient: dL/d
        # pre-activation) as [row][32 points] images; algorithmic bytes = 128 B per 32-row tile per point summed over the
        # 17 GEMMs' operands (field_layout.h wgemm_table: 212 row tiles at ins_num = 13 -> 27.1 KB per point)
        points = n_rays * (2 * N_SAMPLES + N_IMPORTANCE)
        wg_bytes = points * 212 * 128
        t_wg = entry["kernels_ms_per_step"]["field_bwd_weights"] * 1e-3
        entry["roofline_weight_gradients"] = {
            "kernel": "field_wgrad_ring_kernel (nine 256x256 GEMMs) + the small GEMM classes + field_wgrad_reduce_kernel (one event "
                      "around dmn_field_bwd_weights, both networks)", "bound": "hbm", "achieved": wg_bytes / t_wg / 1e9,
            "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": wg_bytes / t_wg / 1e9 / PEAK_HBM_GBS,
            "algorithmic_bytes_per_step": wg_bytes,
            "traffic": None}
        meas = measured_traffic(f"train_rays{n_rays}_ins{INS_NUM}_wgrad")      # counter passes of exactly this step shape
        if meas is not None:
            entry["roofline_weight_gradients"]["traffic"] = meas["hbm_bytes_per_step"]
            entry["roofline_weight_gradients"]["traffic_source"] = meas["source"]
        result[f"forward_{prec}"] = entry
    coarse.load_state_dict(state[0]); fine.load_state_dict(state[1])
    coarse.eval(); fine.eval()
    return result


def measured_traffic(key: str):
    """HBM bytes from the committed rocprofv3 counter passes (profiles/composite_traffic.json), keyed by the exact launch
    / step shape they were taken at; None where no pass exists (the bench then reports ``traffic: null``)."""
    path = os.path.join(ROOT, "profiles", "composite_traffic.json")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        return json.load(f).get(key)


def timed_images(step, steps: int, warmup: int, barrier):
    """`warmup` untimed images, then exactly `steps` images bracketed by barrier + synchronize; per-launch HIP events
    (recorded on the launch stream by ops._timed) are collected over the timed region."""
    for _ in range(warmup):
        step()
    barrier()
    ops.PROFILE = []
    t0 = time.perf_counter()
    out = None
    for _ in range(steps):
        out = step()
    barrier()
    elapsed = time.perf_counter() - t0
    profile, ops.PROFILE = ops.PROFILE, None
    return elapsed, profile, out


def kernel_stats(profile, name):
    rows = [(a.elapsed_time(b) * 1e-3, work) for k, a, b, work in profile if k == name]
    return len(rows), float(sum(r[0] for r in rows)), int(sum(r[1] for r in rows))


def field_roofline(profile, precision: str, elapsed: float):
    """Roofline object of the field kernel from the HIP events of one timed region.

    ``achieved`` is the ALGORITHMIC rate (2 x 693,504 FLOP per sample point / launch time).  ``peak``: the dense MFMA
    peak of the pipe the kernel runs on (MI355X_MICROARCH.md) divided by the matrix-pipe products the arithmetic issues
    per algorithmic product -- 1 for the exact-fp32 MFMA, 3 for the two-way splits, 6 for bf16x6 -- i.e. the ceiling of
    that arithmetic; the same rate against the undivided pipe peak is ``frac_of_pipe_peak_algorithmic``."""
    n_f, t_f, pts_f = kernel_stats(profile, "field_fwd")
    tflops = pts_f * FLOP_PER_POINT / t_f / 1e12
    k = MFMA_PRODUCTS[precision]
    pipe = PEAK_F32_MFMA_TFLOPS if precision == "fp32" else PEAK_16BIT_MFMA_TFLOPS
    name = {"fp32": "field_fwd_kernel<1> (fused encodings + 8x256 field, exact fp32 MFMA v_mfma_f32_32x32x2_f32)",
            "bf16x6": ("field_fwd_x6w_kernel (fused encodings + 8x256 field; fp32 operands split exactly into 3 bf16 planes, "
                       "every fp32 product = 6 v_mfma_f32_16x16x32_bf16 products, fp32 accumulate)" if ops.uses_x6w(INS_NUM)
                       else "field_fwd_x6_kernel (fused encodings + 8x256 field; fp32 operands split exactly into 3 bf16 "
                            "planes, every fp32 product = 6 v_mfma_f32_32x32x16_bf16 products, fp32 accumulate)")}.get(
        precision, f"field_fwd_x3_kernel<{precision}> (fused encodings + 8x256 field; every fp32 product = 3 16-bit MFMA "
                   "products on hi/lo splits, fp32 accumulate)")
    out = {"kernel": name, "bound": "mfma", "achieved": tflops, "peak": pipe / k, "unit": "TFLOP/s", "frac": tflops * k / pipe,
           "traffic": None, "mfma_flop_issued_per_algorithmic_flop": k, "pipe_peak": pipe,
           "frac_of_pipe_peak_algorithmic": tflops / pipe, "launches": n_f, "avg_launch_ms": t_f / max(n_f, 1) * 1e3,
           "algorithmic_flop_per_launch": pts_f * FLOP_PER_POINT / max(n_f, 1), "time_share_of_step": t_f / elapsed,
           # nominal peak is at 2.4 GHz; with every CU on 16-bit MFMA the board power limit holds the clock lower:
           # tools/mfma_peak.hip, profiles/r02a_mfma_shapes.txt
           "peak_sustained_measured": SUSTAINED_MFMA_TFLOPS[precision],
           "mfma_issue_frac_of_sustained": k * tflops / SUSTAINED_MFMA_TFLOPS[precision]}
    if precision != "fp32":     # what the fp32 matrix pipe could do at most for the same algorithmic work
        out["achieved_over_fp32_mfma_peak"] = tflops / PEAK_F32_MFMA_TFLOPS
    return out


def arithmetic_check(nerf, params, device, n_points: int = 4096):
    """(CPU-baseline leg.)  How far each arithmetic is from the exact result: the fine network on `n_points` encoded sample points of the
    benchmark scene, every kernel mode and the CPU oracle's fp32 evaluation (the reference's op sequence) against the
    CPU oracle run in fp64 on the same fp32 parameters.  max |raw - raw64| / max |raw64|."""
    from oracle import dmnerf_cpu as oracle
    pos, view, coarse, fine, args = nerf
    g = torch.Generator().manual_seed(7)
    x = (torch.rand(n_points, 3, generator=g) * 2 - 1) * 6.0
    v = torch.nn.functional.normalize(torch.randn(n_points, 3, generator=g), dim=-1)
    emb = torch.cat([oracle.encode(x, 10), oracle.encode(v, 4)], -1)
    with torch.no_grad():
        want = oracle.field({k: p.double() for k, p in params[1].items()}, emb.double())
        scale = float(want.abs().max())
        res = {"reference_op_sequence_cpu_fp32": float((oracle.field(params[1], emb).double() - want).abs().max()) / scale}
        keep = fine.precision
        for mode in ("fp32", "bf16x6", "fp16x3", "bf16x3"):
            fine.precision = mode
            res[mode] = float((fine(emb.to(device)).cpu().double() - want).abs().max()) / scale
        fine.precision = keep
    res["what"] = (f"max |raw - raw_fp64| / max |raw_fp64| of the fine network on {n_points} encoded points; raw_fp64 = the "
                   "CPU oracle in fp64 on the same fp32 parameters")
    return res


def deviation_stats(got: torch.Tensor, want: torch.Tensor):
    """Per-ray worst deviation relative to max|want| (the north-star's 'relative'): fraction of rays beyond 1e-4,
    99.9th percentile and maximum."""
    got, want = got.double().reshape(got.shape[0], -1), want.double().reshape(want.shape[0], -1)
    per_ray = (got - want).abs().max(-1).values / want.abs().max().clamp_min(1e-30)
    return {"frac_rays_over_1e-4": float((per_ray > 1e-4).double().mean()),
            "p99.9": float(torch.quantile(per_ray, 0.999)), "max": float(per_ray.max())}


def sample_outputs(nerf, K, pose, precision: str, n_rays: int, chunk: int):
    """rgb_fine / depth_fine / ins_fine of the first `n_rays` rays of the image in `chunk`-ray dm_nerf calls."""
    from types import SimpleNamespace
    pos, view, coarse, fine, args = nerf
    args = SimpleNamespace(**{**vars(args), "precision": precision})
    ro, rd = dm.get_rays_k(H, W, K, pose)
    ro, rd = ro.reshape(-1, 3), rd.reshape(-1, 3)
    outs = {"rgb_fine": [], "depth_fine": [], "ins_fine": []}
    with torch.no_grad():
        for s in range(0, n_rays, chunk):
            n = min(chunk, n_rays - s)
            z = dm.z_val_sample(n, args.near, args.far, args.N_samples)
            r = dm.dm_nerf(torch.stack([ro[s:s + n], rd[s:s + n]], 0), pos, view, coarse, fine, z, args)
            for k in outs:
                outs[k].append(r[k])
    return {k: torch.cat(v, 0).cpu() for k, v in outs.items()}


def cpu_baseline(K, pose, params, sample_rays: int, chunk: int, near: float = 4.0, far: float = 15.0):
    """The CPU oracle (port of the reference's PyTorch-CPU path) on a bounded sample of the same workload: the first
    `sample_rays` rays of the image in `chunk`-ray dm_nerf calls (4 chunks of the reference's N_test = 4096 by default,
    BASELINE.md section 4), after a 256-ray warm-up.  Returns rays/s and the sample's outputs (the checker's side of the
    deviation statistics)."""
    from oracle import dmnerf_cpu as oracle
    torch.set_num_threads(host_cores())
    c2w = torch.from_numpy(pose)
    oracle.render_image(H, W, K, c2w, params[0], params[1], near, far, N_SAMPLES, N_IMPORTANCE, chunk=256, max_rays=256)
    rays_o, rays_d = oracle.rays_from_camera(H, W, K, c2w)
    rays_o, rays_d = rays_o.reshape(-1, 3).float(), rays_d.reshape(-1, 3).float()
    outs = {"rgb_fine": [], "depth_fine": [], "ins_fine": []}
    t0 = time.perf_counter()
    with torch.no_grad():
        for s in range(0, sample_rays, chunk):
            n = min(chunk, sample_rays - s)
            r = oracle.render_rays(torch.stack([rays_o[s:s + n], rays_d[s:s + n]], 0), params[0], params[1],
                                   oracle.coarse_depths(n, near, far, N_SAMPLES), n_importance=N_IMPORTANCE)
            for k in outs:
                outs[k].append(r[k])
    dt = time.perf_counter() - t0
    return sample_rays / dt, {k: torch.cat(v, 0) for k, v in outs.items()}


def replica_leg(device, precision: str, chunk: int, steps: int = 3):
    """BASELINE config 3's shape beside the headline (outside its timed region): Replica room_0 geometry -- 640x480,
    ins_num 93 (98-channel raw rows), near 0 / far 6.5 (configs/train/replica/room_0.txt:13-14, loader_replica.py) --
    `steps` images through the chained kernels and through the fused field + compositing kernels, with the roofline of
    composite_wide_kernel (rows wider than 32 channels) from the chained run's HIP events."""
    from types import SimpleNamespace
    from dm_nerf_amd import synthetic
    h, w, ins_num, near, far = 480, 640, 93, 0.0, 6.5
    args = dm.default_args(ins_num=ins_num, N_samples=N_SAMPLES, N_importance=N_IMPORTANCE, N_test=4096, near=near, far=far,
                           device=device)
    pos, view, coarse, fine, args = dm.create_nerf(args)
    coarse.load_state_dict(synthetic.make_params(ins_num, 0))
    fine.load_state_dict(synthetic.make_params(ins_num, 1))
    coarse.eval(); fine.eval()
    hwk, pose = [h, w, synthetic.dmsr_intrinsics(h, w)], synthetic.study_pose(0)
    c_ch = 5 + ins_num
    out = {"workload": f"Replica room_0 shape {w}x{h}, 64+128, ins_num={ins_num} ({c_ch}-channel rows), near {near} / far {far}",
           "precision": precision, "steps": steps, "chunk_rays": chunk}
    images = {}
    for name, fused in (("chained", False), ("fused", True)):
        a = SimpleNamespace(**{**vars(args), "precision": precision, "fused": fused})
        step = lambda: dm.render_image(pos, view, coarse, fine, pose, hwk, a, chunk=chunk)
        elapsed, profile, (rgb, ins) = timed_images(step, steps, 1, torch.cuda.synchronize)
        images[name] = (rgb, ins)
        entry = {"value": steps * h * w / elapsed, "unit": "rays/s", "ms_per_step": elapsed / steps * 1e3}
        n_c, t_c, pts_c = kernel_stats(profile, "composite")
        if n_c:
            per_ray = composite_bytes(1, N_SAMPLES, c_ch) + composite_bytes(1, N_SAMPLES + N_IMPORTANCE, c_ch)
            nbytes = (pts_c // (2 * N_SAMPLES + N_IMPORTANCE)) * per_ray
            entry["roofline_composite"] = {
                "kernel": "composite_wide_kernel (rows straight from global memory, lane per channel)", "bound": "hbm",
                "achieved": nbytes / t_c / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": nbytes / t_c / 1e9 / PEAK_HBM_GBS,
                "traffic": None, "launches": n_c, "avg_launch_ms": t_c / n_c * 1e3, "algorithmic_bytes_per_launch": nbytes / n_c}
            meas = measured_traffic(f"replica_chunk{chunk}")
            if meas is not None:
                entry["roofline_composite"]["traffic"] = meas["hbm_bytes_per_launch"]
                entry["roofline_composite"]["traffic_source"] = meas["source"]
        out[name] = entry
    # the fused kernel runs the network of the 32x32x16 kernels; since round 4 the chained default runs every head width on
    # 16x16x32 (another summation order), so bit identity is checked against the chained path in the fused kernel's variant
    same = ops.save_blob_name(precision)
    ref = images["chained"]
    if same != precision:
        a = SimpleNamespace(**{**vars(args), "precision": same, "fused": False})
        ref = dm.render_image(pos, view, coarse, fine, pose, hwk, a, chunk=chunk)
    out["fused_bit_identical_to_chained"] = bool(torch.equal(ref[0], images["fused"][0]) and torch.equal(ref[1], images["fused"][1]))
    out["fused_identity_checked_against"] = f"the chained path in precision '{same}'"
    out["fused_vs_chained_default_max_abs_rgb"] = float((images["chained"][0] - images["fused"][0]).abs().max())
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--chunk", type=int, default=65536,
                    help="rays per dm_nerf call.  The reference's N_test = 4096 bounds memory on small GPUs; it is a "
                         "tiling parameter (results are partition-invariant bit for bit, tests/test_gpu_parity.py), "
                         "and 65536 saves 2.6 %% of launch ramp and host gaps on the same device (--chunk 4096 for the reference's)")
    ap.add_argument("--cpu-rays", type=int, default=4 * CPU_CHUNK,
                    help="rays of the CPU-baseline sample (0 = skip): 4 chunks of the reference's N_test")
    ap.add_argument("--precision", default=os.environ.get("DMNERF_PRECISION", "bf16x6"),
                    choices=["fp32", "bf16x6", "fp16x3", "bf16x3"],
                    help="arithmetic of the headline `value`: bf16x6 (default) = fp32 operands split exactly into three bf16 "
                         "planes, six bf16 MFMA products per fp32 product, fp32 accumulate (all 24 operand bits enter the "
                         "products); fp32 = exact fp32 MFMA products; fp16x3 / bf16x3 = three 16-bit MFMAs per product on "
                         "two-way splits (16 / 22 operand bits)")
    ap.add_argument("--split-modes", default="fp32,bf16x3,fp16x3",
                    help="the other arithmetics, timed over the same --steps after the headline region and reported beside "
                         "it with their deviation statistics (N = 1 only; '' = none): fp32 lands in `exact_fp32_mode`, the "
                         "two-way splits in `split_precision_modes`")
    ap.add_argument("--no-split-legs", "--no-fp32-leg", dest="no_split", action="store_true",
                    help="same as --split-modes '' (profiling scripts: only the headline mode runs)")
    ap.add_argument("--workload", default="study", choices=["study", "replica"],
                    help="study = BASELINE config 2 (the headline: DM-SR 'study' 800x800, ins_num 13); replica = config 3 "
                         "shape (room_0: 640x480, ins_num 93, near 0 / far 6.5) for a side measurement")
    ap.add_argument("--fused", default="auto", choices=["auto", "on", "off"],
                    help="split-precision modes: field + compositing (+ resampling) fused in one kernel (`raw` never in "
                         "HBM; same bits) or the chained kernels; auto = render_image's rule (fused for rows wider than 32 "
                         "channels)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="N > 1: weak = one image per rank per step (image-sharded, no data-path collective); strong = ONE "
                         "image per step, its flat ray range cut into N contiguous blocks (parallel.ray_shard), every rank "
                         "renders its block and the finished maps are all-gathered (RCCL) inside the timed region")
    ap.add_argument("--no-side-legs", action="store_true",
                    help="skip the chunk-4096 and Replica-shape measurements after the timed region (N = 1 only)")
    ap.add_argument("--no-train-leg", action="store_true",
                    help="skip the extra training-step measurement after the timed region (N = 1 only)")
    ns = ap.parse_args()

    global H, W, INS_NUM, MACS_PER_POINT, FLOP_PER_POINT
    near, far, scene = 4.0, 15.0, "DM-SR 'study' 800x800"
    if ns.workload == "replica":            # configs/train/replica/room_0.txt:13-14, loader_replica.py
        H, W, INS_NUM, near, far, scene = 480, 640, 93, 0.0, 6.5, "Replica room_0 shape 640x480"
        MACS_PER_POINT = 693_504 + (93 - 13) * 129          # ins_linear grows by 80 rows of 128 weights + bias
        FLOP_PER_POINT = 2 * MACS_PER_POINT

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != ns.gpus:
        raise SystemExit(f"--gpus {ns.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run for N > 1")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (the HIP path has no CPU fallback)")
    # one rank per GPU; DMNERF_DIST_BACKEND=gloo lets several ranks share one GPU for a rehearsal of the N > 1 path
    backend = os.environ.get("DMNERF_DIST_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=device)
        else:
            dist.init_process_group(backend=backend)

    dm.load_library()
    nerf, K, poses, params = synthetic_scene(device, near, far)
    pos, view, coarse, fine, args = nerf
    pose = poses[rank % len(poses)]
    hwk = [H, W, K]

    strong = ns.scaling == "strong" and world > 1
    if strong:
        pose = poses[0]                      # every rank works on the same image

    def make_step(precision, chunk=None, **kw):
        from types import SimpleNamespace
        a = SimpleNamespace(**{**vars(args), "precision": precision, "fused": {"auto": None, "on": True, "off": False}[ns.fused]})
        return lambda: dm.render_image(pos, view, coarse, fine, pose, hwk, a, chunk=chunk or ns.chunk, **kw)

    def make_strong_step(precision):
        """One image, ray-sharded: this rank's contiguous block of the flat ray range, then the all-gather of the maps."""
        from dm_nerf_amd.parallel import gather_rays, ray_shard
        block = ray_shard(H * W, rank, world)
        local = make_step(precision, ray_range=block)

        def step():
            rgb_l, ins_l = local()
            rgb_g = gather_rays(rgb_l, H * W)
            ins_g = gather_rays(ins_l, H * W)
            return rgb_g.reshape(H, W, 3), ins_g.reshape(H, W, ins_g.shape[-1])
        return step

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- the timed region: --warmup untimed images, then exactly --steps images
    headline_step = make_strong_step(ns.precision) if strong else make_step(ns.precision)
    elapsed, profile, (rgb, ins) = timed_images(headline_step, ns.steps, ns.warmup, barrier)
    if dist is not None:
        from dm_nerf_amd.parallel import max_over_ranks
        elapsed = max_over_ranks(elapsed, device)
    sharded_equal = None
    if strong and rank == 0:                 # outside the timed region: the same image on one rank, bit for bit
        rgb_1, ins_1 = make_step(ns.precision)()
        sharded_equal = bool(torch.equal(rgb_1, rgb) and torch.equal(ins_1, ins))

    if rank != 0:
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    rays_total = (1 if strong else world) * ns.steps * H * W
    value = rays_total / elapsed
    n_c, t_c, pts_c = kernel_stats(profile, "composite")
    c_ch = 5 + INS_NUM
    # composite launches alternate S=64 and S=192 with the same ray count; bytes by sample total
    comp_bytes = (pts_c // (N_SAMPLES + N_SAMPLES + N_IMPORTANCE)) * (
        composite_bytes(1, N_SAMPLES, c_ch) + composite_bytes(1, N_SAMPLES + N_IMPORTANCE, c_ch))
    comp_gbs = comp_bytes / t_c / 1e9 if t_c > 0 else 0.0      # fused modes have no stand-alone compositing launches
    dtype_names = {"fp32": "f32", "fp16x3": "f16x3 (hi/lo split, f32 accumulate)",
                   "bf16x3": "bf16x3 (hi/lo split, f32 accumulate)",
                   "bf16x6": "f32 operands as exact 3-way bf16 splits, 6 bf16 MFMA products per product, f32 accumulate "
                             "(bf16x6)"}

    out = {
        "metric": f"rays/sec, {'DM-SR study 800x800' if ns.workload == 'study' else scene}, 64+128 samples (PSNR vs "
                  "reference path in `psnr_vs_cpu_oracle_db`)",
        "value": value, "unit": "rays/s", "n_gpus": world, "steps": ns.steps, "warmup": ns.warmup,
        "ms_per_step": elapsed / ns.steps * 1e3, "higher_is_better": True, "scaling": "strong" if strong else "weak",
        "vs_baseline": None, "dtype": dtype_names[ns.precision], "data": "synthetic",
        "config": {"workload": f"{scene}, 64+128 hierarchical, 8x256 field, RGB+sigma+instance head "
                               f"(ins_num={INS_NUM}), " + ("1 image per step, ray-sharded over the GPUs" if strong
                                                           else "1 image per GPU per step"), "chunk_rays": ns.chunk,
                   "precision": ns.precision,
                   "parallelism": (f"one image ray-sharded x{world} (contiguous blocks), maps all-gathered in the timed region"
                                   if strong else f"ray/image-sharded x{world}, no collective")},
        "roofline": field_roofline(profile, ns.precision, elapsed),
        "roofline_composite": {"kernel": ("composite_wide_kernel (rows straight from global memory, lane per channel)"
                                          if c_ch > 32 else "composite_kernel (LDS-staged slabs + wave product scan)"),
                               "bound": "hbm", "achieved": comp_gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                               "frac": comp_gbs / PEAK_HBM_GBS, "traffic": None, "launches": n_c,
                               "avg_launch_ms": t_c / max(n_c, 1) * 1e3, "time_share_of_step": t_c / elapsed,
                               "algorithmic_bytes_per_launch": comp_bytes / max(n_c, 1)},
    }

    if strong:
        out["sharded_image_bit_equal_to_one_rank_image"] = sharded_equal
        if not sharded_equal:
            raise SystemExit("strong scaling: the ray-sharded image differs from the one-rank image")

    traffic_file = os.path.join(ROOT, "profiles", "composite_traffic.json")
    if os.path.exists(traffic_file):        # HBM bytes per composite launch from rocprofv3 PMC passes (tools/pmc_composite.sh)
        meas = json.load(open(traffic_file)).get(f"{ns.workload}_chunk{ns.chunk}")
        if meas is not None:                # measured at exactly this launch size (rays per launch, S mix, C)
            out["roofline_composite"]["traffic"] = meas["hbm_bytes_per_launch"]
            out["roofline_composite"]["traffic_source"] = meas["source"]
        meas = json.load(open(traffic_file)).get(f"{ns.workload}_chunk{ns.chunk}_field_{ns.precision}")
        if meas is not None and INS_NUM == 13:      # the headline kernel's own HBM bytes per (average) launch
            out["roofline"]["traffic"] = meas["hbm_bytes_per_launch"]
            out["roofline"]["traffic_source"] = meas["source"]
            out["roofline"]["algorithmic_bytes_per_launch"] = meas["algorithmic_bytes_per_launch"]

    # ---- beside the headline (N = 1 only, outside its timed region): the other arithmetics over the same --steps
    modes = [m for m in ns.split_modes.split(",") if m and m != ns.precision] if world == 1 and not ns.no_split else []
    images = {ns.precision: rgb}
    exact_rgb = rgb if ns.precision == "fp32" else None
    if "fp32" in modes:         # first: the whole-image deviation of every other mode is taken against this render
        e_m, prof_m, (exact_rgb, _) = timed_images(make_step("fp32"), ns.steps, min(ns.warmup, 2) or 1, barrier)
        images["fp32"] = exact_rgb
        out["exact_fp32_mode"] = {"value": ns.steps * H * W / e_m, "unit": "rays/s", "steps": ns.steps,
                                  "ms_per_step": e_m / ns.steps * 1e3, "dtype": dtype_names["fp32"],
                                  "roofline": field_roofline(prof_m, "fp32", e_m)}
        meas = measured_traffic(f"{ns.workload}_chunk{ns.chunk}_field_fp32")
        if meas is not None and INS_NUM == 13:
            out["exact_fp32_mode"]["roofline"]["traffic"] = meas["hbm_bytes_per_launch"]
            out["exact_fp32_mode"]["roofline"]["traffic_source"] = meas["source"]
        out["headline_over_exact_fp32_mode"] = value / out["exact_fp32_mode"]["value"]
    if exact_rgb is not None and ns.precision != "fp32":
        out["whole_image_vs_fp32_mode"] = deviation_stats(rgb.reshape(-1, 3), exact_rgb.reshape(-1, 3))
    if world == 1 and not ns.no_split and ns.precision != "fp32":
        # the headline arithmetic with the coarse pass reduced to density (render_image(image_only=True)): bit-identical
        # images, the coarse network's colour / instance heads are not evaluated.  Reported beside, never as `value`.
        only = make_step(ns.precision, image_only=True)
        rgb_only, _ = only()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        only()
        torch.cuda.synchronize()
        dt_only = time.perf_counter() - t1
        out["image_only_coarse_pass"] = {"value": H * W / dt_only, "unit": "rays/s", "ms_per_step": dt_only * 1e3,
                                         "identical_image": bool(torch.equal(rgb_only, rgb))}
    modes = [m for m in modes if m != "fp32"]
    if modes:
        out["split_precision_modes"] = {}
    for mode in modes:
        e_m, prof_m, (rgb_m, _) = timed_images(make_step(mode), ns.steps, min(ns.warmup, 2) or 1, barrier)
        images[mode] = rgb_m
        entry = {"value": ns.steps * H * W / e_m, "unit": "rays/s", "steps": ns.steps, "ms_per_step": e_m / ns.steps * 1e3,
                 "dtype": dtype_names[mode], "roofline": field_roofline(prof_m, mode, e_m)}
        if exact_rgb is not None:           # whole image against the exact-arithmetic render of the same pose
            entry["whole_image_vs_fp32_mode"] = deviation_stats(rgb_m.reshape(-1, 3), exact_rgb.reshape(-1, 3))
        # the same image loop with the coarse pass reduced to density (render_image(image_only=True)): bit-identical images,
        # the coarse network's colour / instance heads are not evaluated.  Reported beside, never as `value`.
        only = make_step(mode, image_only=True)
        rgb_only, _ = only()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        only()
        torch.cuda.synchronize()
        dt_only = time.perf_counter() - t1
        entry["image_only_coarse_pass"] = {"value": H * W / dt_only, "unit": "rays/s", "ms_per_step": dt_only * 1e3,
                                           "identical_image": bool(torch.equal(rgb_only, rgb_m))}
        out["split_precision_modes"][mode] = entry
    if world == 1 and not ns.no_side_legs and ns.workload == "study":
        # the reference's own chunk size (N_test = 4096, configs/test/dmsr/study.txt:10; networks/tester.py:62) over the
        # same image: 157 dm_nerf calls per image instead of 10
        n4 = max(1, min(ns.steps, 5))
        e4, prof4, (rgb4, _) = timed_images(make_step(ns.precision, chunk=CPU_CHUNK), n4, 1, barrier)
        out["chunk_4096"] = {"value": n4 * H * W / e4, "unit": "rays/s", "steps": n4, "ms_per_step": e4 / n4 * 1e3,
                             "chunk_rays": CPU_CHUNK, "ratio_to_headline": (n4 * H * W / e4) / value,
                             "identical_image": bool(torch.equal(rgb4, rgb)),
                             "field_kernel_avg_launch_ms": field_roofline(prof4, ns.precision, e4)["avg_launch_ms"]}
        out["replica_shape"] = replica_leg(device, ns.precision, ns.chunk)
    if world == 1 and not ns.no_train_leg:
        out["training_step"] = training_leg(nerf, K, poses[0], device)

    if ns.cpu_rays > 0 and world == 1:       # the CPU baseline is timed on rank 0 at N = 1 only
        cpu_rps, cpu_out = cpu_baseline(K, pose, params, ns.cpu_rays, CPU_CHUNK, near, far)
        got = rgb.reshape(-1, 3)[: ns.cpu_rays].cpu()
        mse = float(torch.mean((got - cpu_out["rgb_fine"]) ** 2))
        out["psnr_vs_cpu_oracle_db"] = float(-10.0 * np.log10(max(mse, 1e-20)))
        out["max_abs_err_rgb_vs_cpu_oracle"] = float((got - cpu_out["rgb_fine"]).abs().max())
        out["cpu_baseline"] = {
            "value": cpu_rps, "unit": "rays/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"first {ns.cpu_rays} rays of the same image in {CPU_CHUNK}-ray chunks "
                      f"({ns.cpu_rays // CPU_CHUNK} chunks) through oracle/dmnerf_cpu.py (PyTorch CPU fp32, same op "
                      "sequence as the reference), after a 256-ray warm-up"}
        out["gpu_over_cpu"] = value / cpu_rps
        # part of the CPU leg: the oracle in fp64 as the checker of every arithmetic (and of its own fp32 evaluation)
        out["arithmetic_check"] = arithmetic_check(nerf, params, device)
        # deviation of every mode from the CPU oracle on the sample rays, the same chunking on both sides
        # (relative = per-ray worst element over max|oracle|; the oracle itself moves by 2e-4 against an fp64 run of the
        # reference behind the sampler)
        dev = {}
        for mode in [ns.precision] + (["fp32"] if "exact_fp32_mode" in out else []) + modes:
            gpu_out = sample_outputs(nerf, K, pose, mode, ns.cpu_rays, CPU_CHUNK)
            dev[mode] = {k: deviation_stats(gpu_out[k], cpu_out[k]) for k in ("rgb_fine", "depth_fine", "ins_fine")}
        out["deviation_vs_cpu_oracle"] = dev
    print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
