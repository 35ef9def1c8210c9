"""Eager chunk loop against tester.GraphedImage at the reference's chunk size (N_test = 4096) on the 800x800 image: host
calls and wall time per image.  GPU box:  python tools/graph_image_check.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dm_nerf_amd as dm
from dm_nerf_amd import synthetic
dev = torch.device("cuda:0")
args = dm.default_args(ins_num=13, device=dev)
pos, view, coarse, fine, args = dm.create_nerf(args)
coarse.load_state_dict(synthetic.make_params(13, 0)); fine.load_state_dict(synthetic.make_params(13, 1))
coarse.eval(); fine.eval()
hwk = [800, 800, synthetic.dmsr_intrinsics(800, 800)]
for chunk in (4096, 65536):
    renderer = dm.GraphedImage(pos, view, coarse, fine, hwk, args, chunk=chunk)
    for name, fn in (("eager", lambda p: dm.render_image(pos, view, coarse, fine, p, hwk, args, chunk=chunk)), ("graph", renderer)):
        fn(synthetic.study_pose(0)); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(3):
            out = fn(synthetic.study_pose(i))
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        print(f"chunk {chunk:6d} {name}: {dt * 1e3:8.2f} ms per image, {640000 / dt / 1e3:7.1f} k rays/s")
