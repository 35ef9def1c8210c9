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
        slowest = max_over_ranks(float(rank + 1), torch.device("cpu"))
        if rank == 0:
            np.savez(result_path, full=full.numpy(), slowest=slowest)
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
