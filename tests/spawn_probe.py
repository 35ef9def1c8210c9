"""Stand-in for bench.py's launch prologue on a machine without a GPU (tests/test_parallel.py): with ``--gpus N`` and no
launcher in the environment it starts itself as N ranks under torch.distributed.run through dm_nerf_amd/launch.py -- the
very code path of ``python bench.py --gpus N`` -- and each rank joins a gloo group; rank 0 prints one JSON line."""
import importlib.util
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("_dmn_launch", os.path.join(ROOT, "dm_nerf_amd", "launch.py"))
launch = importlib.util.module_from_spec(spec)
spec.loader.exec_module(launch)

if __name__ == "__main__":
    if launch.needs_self_launch(sys.argv[1:]):
        assert "torch" not in sys.modules, "the launching parent must not have imported torch"
        sys.exit(launch.self_launch(os.path.abspath(__file__), sys.argv[1:]))
    import torch
    import torch.distributed as dist
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    assert world == launch.requested_gpus(sys.argv[1:])
    if "--fail-on-rank" in sys.argv and rank == int(sys.argv[sys.argv.index("--fail-on-rank") + 1]):
        sys.exit(3)
    if world > 1:
        dist.init_process_group(backend="gloo")
        t = torch.tensor([float(rank + 1)])
        dist.all_reduce(t)
        total = float(t)
        dist.barrier()
        dist.destroy_process_group()
    else:
        total = 1.0
    if rank == 0:
        print(json.dumps({"n_gpus": world, "sum_of_ranks_plus_one": total, "master_addr": os.environ.get("MASTER_ADDR")}), flush=True)
