"""Self-launch of the one-process-per-GPU benchmarks: ``python bench.py --gpus N`` (no ``WORLD_SIZE`` in the
environment) re-runs itself under ``python -m torch.distributed.run`` with N local ranks and relays what rank 0 prints.

This module imports neither torch nor the HIP library: the parent must not initialise the GPU (a process that has may
not be replaced or fork GPU children safely on this pool), it only starts the launcher as a child process, waits, and
exits with its status."""
from __future__ import annotations

import os
import socket
import subprocess
import sys
from typing import List, Optional


def requested_gpus(argv: List[str]) -> int:
    """Value of ``--gpus`` in ``argv`` (``--gpus N`` or ``--gpus=N``), 1 when absent."""
    for i, a in enumerate(argv):
        if a == "--gpus" and i + 1 < len(argv):
            return int(argv[i + 1])
        if a.startswith("--gpus="):
            return int(a.split("=", 1)[1])
    return 1


def needs_self_launch(argv: List[str], environ=os.environ) -> bool:
    """True for ``--gpus N`` with N > 1 when no launcher has set up the ranks yet."""
    return requested_gpus(argv) > 1 and "WORLD_SIZE" not in environ and "RANK" not in environ


def free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launcher_command(script: str, argv: List[str], n: int, port: Optional[int] = None) -> List[str]:
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
            "--master-addr", "127.0.0.1", "--master-port", str(port or free_port()), script, *argv]


def self_launch(script: str, argv: List[str]) -> int:
    """Run ``script argv`` as ``--gpus`` ranks under torch.distributed.run; stdout / stderr pass through (rank 0 prints
    the one JSON line).  Returns the launcher's exit status."""
    n = requested_gpus(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL needs on this host driver
    env.setdefault("OMP_NUM_THREADS", "1")
    proc = subprocess.run(launcher_command(script, argv, n), env=env)
    return proc.returncode
