"""N > 1 host logic on CPU: two gloo ranks shard one image's rays, render their blocks (with the
CPU oracle standing in for the device, there is no GPU here), gather, and must reproduce the
unsharded render: the gather itself exactly (a ray-index column travels with the pixels), the pixel
values to 1e-5 (CPU GEMM blocking depends on the batch shape; on the GPU the HIP path is
partition-invariant bit for bit, tests/test_gpu_parity.py::test_full_size_chunk_properties)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import cases
from oracle import dmnerf_cpu as oracle
from dm_nerf_amd.parallel import allreduce_gradients, gather_rays, max_over_ranks, pose_shard, ray_shard

H, W, CHUNK = 10, 12, 50


def test_ray_shard_tiles_the_range():
    for n in (0, 1, 7, 640000, 640001):
        for world in (1, 2, 3, 8):
            blocks = [ray_shard(n, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(blocks, blocks[1:]))
            sizes = [hi - lo for lo, hi in blocks]
            assert max(sizes) - min(sizes) <= 1
    assert ray_shard(640000, 3, 8) == (240000, 320000)
    assert pose_shard(10, 1, 4) == [1, 5, 9]
    with pytest.raises(ValueError):
        ray_shard(10, 2, 2)


def _render_block(lo, hi):
    K, c2w = cases.study_intrinsics(H, W), torch.from_numpy(cases.study_c2w())
    rays_o, rays_d = oracle.rays_from_camera(H, W, K, c2w)
    rays_o, rays_d = rays_o.reshape(-1, 3)[lo:hi], rays_d.reshape(-1, 3)[lo:hi]
    coarse, fine = oracle.make_params(13, 0), oracle.make_params(13, 1)
    out = []
    with torch.no_grad():
        for s in range(0, hi - lo, CHUNK):
            n = min(CHUNK, hi - lo - s)
            z = oracle.coarse_depths(n, cases.NEAR, cases.FAR, 64)
            r = oracle.render_rays(torch.stack([rays_o[s:s + n], rays_d[s:s + n]], 0), coarse, fine, z)
            idx = torch.arange(lo + s, lo + s + n, dtype=torch.float32)[:, None]
            out.append(torch.cat([idx, r["rgb_fine"], r["depth_fine"][:, None], r["ins_fine"]], -1))
    return torch.cat(out, 0) if out else torch.zeros(0, 18)


def _worker(rank, world, port, result_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = ray_shard(H * W, rank, world)
        local = _render_block(lo, hi)
        full = gather_rays(local, H * W)
        a, b = ray_shard(7, rank, world)                      # ragged blocks (4 + 3 rays): the padded path
        ragged = gather_rays(torch.arange(a, b, dtype=torch.float32).reshape(-1, 1) * 2.0, 7)
        slowest = max_over_ranks(float(rank + 1), torch.device("cpu"))
        if rank == 0:
            np.savez(result_path, full=full.numpy(), slowest=slowest, ragged=ragged.numpy())
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_rank_sharded_render_equals_single_rank(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    result = str(tmp_path / "gathered.npz")
    mp.spawn(_worker, args=(2, port, result), nprocs=2, join=True)
    got = np.load(result)
    torch.set_num_threads(4)
    want = _render_block(0, H * W).numpy()
    np.testing.assert_array_equal(got["full"][:, 0], np.arange(H * W, dtype=np.float32))   # placement exact
    np.testing.assert_allclose(got["full"][:, 1:], want[:, 1:], rtol=0, atol=1e-5)
    assert float(got["slowest"]) == 2.0
    np.testing.assert_array_equal(got["ragged"][:, 0], 2.0 * np.arange(7, dtype=np.float32))


def _grad_worker(rank, world, port, result_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        nets = [torch.nn.Linear(5, 3), torch.nn.Sequential(torch.nn.Linear(3, 4), torch.nn.Linear(4, 2))]
        x = torch.full((7, 5), float(rank + 1))                   # rank-dependent data -> rank-dependent grads
        nets[1](nets[0](x)).sum().backward()
        local = [p.grad.clone() for m in nets for p in m.parameters()]
        n = allreduce_gradients(nets)
        after = [p.grad.clone() for m in nets for p in m.parameters()]
        torch.save({"n": n, "local": local, "after": after}, f"{result_path}.{rank}")
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_flat_gradient_allreduce_two_ranks(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    base = str(tmp_path / "grads")
    mp.spawn(_grad_worker, args=(2, port, base), nprocs=2, join=True)
    r0, r1 = torch.load(base + ".0"), torch.load(base + ".1")
    assert r0["n"] == r1["n"] == sum(g.numel() for g in r0["local"])
    for a0, a1, l0, l1 in zip(r0["after"], r1["after"], r0["local"], r1["local"]):
        torch.testing.assert_close(a0, a1, rtol=0, atol=0)          # every rank holds the same reduced gradient
        torch.testing.assert_close(a0, (l0 + l1) / 2)               # the mean of the per-rank gradients


# ---- batch-exact sharded loss: statistics all-reduce + shard-weighted gradient SUM --------------------------------
def _toy_loss(pred, labels, n_classes, reduce, share):
    """Same shape of computation as the sharded instance / colour losses: class-wise sums of a function of the
    predictions (all-reduced), a loss that is NON-linear in those sums and normalised by the batch's ray count, plus a
    mean-type term weighted by the shard's share."""
    onehot = torch.nn.functional.one_hot(labels, n_classes).float()
    stats = onehot.t() @ torch.log(pred + 1e-3)                       # [classes, K] sums over rays
    count = onehot.sum(0)
    n = torch.tensor([float(pred.shape[0])])
    if reduce is not None:
        stats, count, n = reduce("stats", stats), reduce("count", count), reduce("n", n)
    matched = (stats / n).exp().sum() / (count > 0).sum()
    return matched + share * (pred ** 2).mean()


def _exact_worker(rank, world, port, result_path):
    from dm_nerf_amd.parallel import GradBucket, ShardReducer
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        net = torch.nn.Sequential(torch.nn.Linear(6, 8), torch.nn.Sigmoid(), torch.nn.Linear(8, 4), torch.nn.Sigmoid())
        g = torch.Generator().manual_seed(1)
        x, labels = torch.rand(37, 6, generator=g), torch.randint(0, 5, (37,), generator=g)
        lo, hi = ray_shard(37, rank, world)                            # unequal shards: 19 + 18
        bucket = GradBucket([net])
        loss = _toy_loss(net(x[lo:hi]), labels[lo:hi], 5, ShardReducer(), (hi - lo) / 37)
        loss.backward()
        n = bucket.allreduce(scale=1.0)
        assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(bucket.params, bucket.views))
        # a second step reuses the buffer: grads dropped (set_to_none), backward again, same result
        first = bucket.flat.clone()
        for p in net.parameters():
            p.grad = None
        _toy_loss(net(x[lo:hi]), labels[lo:hi], 5, ShardReducer(), (hi - lo) / 37).backward()
        bucket.allreduce(scale=1.0)
        torch.save({"n": n, "flat": bucket.flat.clone(), "first": first, "loss": loss.detach()}, f"{result_path}.{rank}")
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_sharded_statistics_give_the_unsharded_loss_and_gradient(tmp_path):
    """Two gloo ranks with unequal shards: ShardReducer (sum of the batch statistics, identity backward) + shard-
    weighted mean term + GradBucket SUM reproduce the single-process loss and gradient of the whole batch."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    base = str(tmp_path / "exact")
    mp.spawn(_exact_worker, args=(2, port, base), nprocs=2, join=True)
    r0, r1 = torch.load(base + ".0"), torch.load(base + ".1")
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(6, 8), torch.nn.Sigmoid(), torch.nn.Linear(8, 4), torch.nn.Sigmoid())
    g = torch.Generator().manual_seed(1)
    x, labels = torch.rand(37, 6, generator=g), torch.randint(0, 5, (37,), generator=g)
    loss = _toy_loss(net(x), labels, 5, None, 1.0)
    loss.backward()
    want = torch.cat([p.grad.reshape(-1) for p in net.parameters()])
    assert r0["n"] == want.numel()
    torch.testing.assert_close(r0["flat"], r1["flat"], rtol=0, atol=0)          # replicas agree bit for bit
    torch.testing.assert_close(r0["flat"], r0["first"], rtol=0, atol=0)         # buffer reuse is clean
    torch.testing.assert_close(r0["flat"], want, rtol=1e-5, atol=1e-7)
    # the matched term is replicated, the mean term is a share: loss_0 + loss_1 - matched = whole-batch loss
    matched = loss - (net(x) ** 2).mean()
    torch.testing.assert_close(r0["loss"] + r1["loss"] - matched.detach(), loss.detach(), rtol=1e-5, atol=1e-7)


def test_bench_starts_its_own_ranks_when_no_launcher_did(tmp_path):
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment: the parent starts N ranks under
    torch.distributed.run BEFORE importing torch and exits with their status (dm_nerf_amd/launch.py).  The probe script
    runs that exact prologue with gloo ranks on the CPU."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    probe = os.path.join(root, "tests", "spawn_probe.py")
    res = subprocess.run([sys.executable, probe, "--gpus", "2"], env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout
    got = json.loads(lines[0])
    assert got == {"n_gpus": 2, "sum_of_ranks_plus_one": 3.0, "master_addr": "127.0.0.1"}
    # a failing rank is the parent's exit status, not a hang and not rc 0
    res = subprocess.run([sys.executable, probe, "--gpus=2", "--fail-on-rank", "1"], env=env, capture_output=True, text=True,
                         timeout=300)
    assert res.returncode != 0
    # one rank needs no launcher
    res = subprocess.run([sys.executable, probe], env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and json.loads(res.stdout.strip().splitlines()[-1])["n_gpus"] == 1
    # bench.py and tools/bench_train.py run the same prologue before they import torch
    for script in ("bench.py", os.path.join("tools", "bench_train.py")):
        src = open(os.path.join(root, script)).read()
        assert "_self_launch_if_needed()" in src
        assert src.index("_self_launch_if_needed()\n\n") < src.index("\nimport torch"), script


def test_launch_helpers():
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("_dmn_launch", os.path.join(root, "dm_nerf_amd", "launch.py"))
    launch = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(launch)
    assert launch.requested_gpus(["--steps", "3"]) == 1
    assert launch.requested_gpus(["--gpus", "4", "--steps", "3"]) == 4 and launch.requested_gpus(["--gpus=8"]) == 8
    assert launch.needs_self_launch(["--gpus", "2"], environ={})
    assert not launch.needs_self_launch(["--gpus", "2"], environ={"WORLD_SIZE": "2", "RANK": "0"})
    assert not launch.needs_self_launch(["--gpus", "1"], environ={})
    cmd = launch.launcher_command("bench.py", ["--gpus", "2", "--steps", "1"], 2, port=29512)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=2" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-5:] == ["bench.py", "--gpus", "2", "--steps", "1"]
