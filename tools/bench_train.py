#!/usr/bin/env python3
"""Training-step timing (BASELINE config 4): N_rand rays per step, stochastic path, Adam.  Prints one JSON line.
Not the headline bench (bench.py is).

    python tools/bench_train.py [--rays 4096 --steps 5 --precision bf16x6|fp32|bf16x3 --instance --penalize]
    python tools/bench_train.py --gpus N        (starts its own N ranks; or under torch.distributed.run --nproc-per-node N)

With N ranks the rays of a step are sharded (strong scaling: --rays is the global batch), every rank runs forward and
the hand-written backward on its shard, the loss heads all-reduce their batch statistics (a few hundred bytes:
class statistics of the instance matching, mask sums of the penalizer) so that the step optimises the loss of the WHOLE
batch, and ONE in-place all-reduce (RCCL over xGMI; 5.57 MB) over the persistent flat gradient buffer sums the
shard-weighted gradients before Adam.  DMNERF_DIST_BACKEND=gloo rehearses the N > 1 path on one GPU."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _self_launch_if_needed():
    """`--gpus N` without a launcher: start N ranks under torch.distributed.run as a child process (before torch is
    imported: the parent never touches the GPU) and exit with its status; dm_nerf_amd/launch.py, loaded by path."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("_dmn_launch", os.path.join(ROOT, "dm_nerf_amd", "launch.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    if mod.needs_self_launch(sys.argv[1:]):
        sys.exit(mod.self_launch(os.path.abspath(__file__), sys.argv[1:]))


if __name__ == "__main__":
    _self_launch_if_needed()

import torch  # noqa: E402
import dm_nerf_amd as dm  # noqa: E402
from dm_nerf_amd import parallel, synthetic, trainer  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--gpus", type=int, default=1)
ap.add_argument("--rays", type=int, default=4096, help="rays per optimisation step over all ranks (N_rand)")
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--precision", default="bf16x6", choices=["fp32", "bf16x6", "bf16x3", "fp16x3"])
ap.add_argument("--instance", action="store_true", help="add the instance-matching loss on random labels")
ap.add_argument("--assign", default="device", choices=["host", "device"],
                help="Hungarian step of the instance loss: scipy on the host (one wait for the device per step) or the kernel")
ap.add_argument("--separate-ins", action="store_true", help="evaluate the two instance maps in two calls (two waits for the device)")
ap.add_argument("--penalize", action="store_true", help="add the emptiness penalizer (configs: tolerance = deta_w = 0.05)")
ns = ap.parse_args()

world = int(os.environ.get("WORLD_SIZE", "1"))
rank = int(os.environ.get("RANK", "0"))
local_rank = int(os.environ.get("LOCAL_RANK", "0"))
if world != ns.gpus:
    raise SystemExit(f"--gpus {ns.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run for N > 1")
backend = os.environ.get("DMNERF_DIST_BACKEND", "nccl")
dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
torch.cuda.set_device(dev_index)
dev = torch.device("cuda", dev_index)
dist = None
if world > 1:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(backend=backend, **({"device_id": dev} if backend == "nccl" else {}))

args = dm.default_args(ins_num=13, perturb=1.0, is_train=True, precision=ns.precision, device=dev)
pos, view, coarse, fine, args = dm.create_nerf(args)
coarse.load_state_dict(synthetic.make_params(13, 0))       # every rank starts from the same weights
fine.load_state_dict(synthetic.make_params(13, 1))
K, c2w = synthetic.dmsr_intrinsics(800, 800), synthetic.study_pose(1)
ro, rd = dm.get_rays_k(800, 800, K, c2w)
g = torch.Generator(device=dev).manual_seed(7)              # same draws on every rank, each takes its block
pick = torch.randperm(640000, device=dev, generator=g)[: ns.rays]
target = torch.rand(ns.rays, 3, device=dev, generator=g)
labels = torch.randint(0, 13, (ns.rays,), device=dev, generator=g)
lo, hi = parallel.ray_shard(ns.rays, rank, world)
pick, target, labels = pick[lo:hi], target[lo:hi], labels[lo:hi]
rays = torch.stack([ro.reshape(-1, 3)[pick], rd.reshape(-1, 3)[pick]], 0)
z = dm.z_val_sample(hi - lo, args.near, args.far, args.N_samples)
opt = trainer.make_optimizer(coarse, fine)
args.tolerance, args.deta_w = 0.05, 0.05
reducer = parallel.ShardReducer() if world > 1 else None
bucket = parallel.GradBucket([coarse, fine])                # the weight-gradient kernels write straight into it
ins = trainer.instance_loss(labels, 13, reduce=reducer, assign=ns.assign) if ns.instance else None
if ins is not None and ns.separate_ins and hasattr(ins, "many"):
    del ins.many
pen = (lambda raw, zz, depth: dm.ins_penalizer(raw, zz, depth, rays[1], args, reduce=reducer)) if ns.penalize else None
reduce_grads = (lambda models: bucket.allreduce()) if world > 1 else None       # SUM of shard-weighted gradients


def step(i):
    return trainer.train_step(i, rays, target, (pos, view, coarse, fine, args), z, opt, ins_loss=ins, penalizer=pen,
                              allreduce=reduce_grads, shard=(hi - lo, ns.rays) if world > 1 else None, report=False)


def barrier():
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()


step(0)
barrier()
t0 = time.perf_counter()
for i in range(ns.steps):
    step(i + 1)
barrier()
dt = time.perf_counter() - t0
in_sync = True
if dist is not None:
    dt = parallel.max_over_ranks(dt, dev)
    # the replicas must still agree bit for bit after the averaged updates
    w = torch.cat([p.detach().reshape(-1) for p in coarse.parameters()])
    top = w.clone()
    dist.all_reduce(top, op=dist.ReduceOp.MAX)
    flag = torch.tensor([1.0 if torch.equal(top, w) else 0.0], device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    in_sync = bool(flag.item() == 1.0)
dt /= ns.steps
if rank == 0:
    print(json.dumps({"metric": "training rays/s (HIP field fwd+save, HIP composite bwd, HIP field backward-data + "
                                "weight-gradient kernels, flat gradient all-reduce, Adam)",
                      "n_gpus": world, "rays_per_step": ns.rays, "rays_per_rank": hi - lo, "ms_per_step": dt * 1e3,
                      "value": ns.rays / dt, "unit": "rays/s", "scaling": "strong", "precision_forward": ns.precision,
                      "emptiness_penalizer": bool(ns.penalize), "instance_loss": bool(ns.instance),
                      "replicas_in_sync": in_sync, "peak_mem_gb": torch.cuda.max_memory_allocated() / 2 ** 30}))
if dist is not None:
    dist.barrier()
    dist.destroy_process_group()
