"""One process per GPU, started from a plain ``python script.py --gpus N``.

Replaces the reference's process start-up (``mp.spawn(main_worker, nprocs=ngpus)`` with a
hard-coded ``tcp://127.0.0.1:2345`` rendezvous, main.py:100-132): the parent -- which must not
have touched the GPU, since a process that has initialised HIP is neither forked nor re-executed
on this platform -- starts N fresh interpreters of the same script with RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_ADDR / MASTER_PORT set (the variables ``torch.distributed.run`` would set, so a
script cannot tell the two launchers apart), relays rank 0's stdout and fails if any rank fails.
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys
import time
from typing import List, Optional, Sequence


def free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def under_launcher() -> bool:
    """True inside a rank started by ``torch.distributed.run`` or by ``spawn_ranks``."""
    return "WORLD_SIZE" in os.environ and "RANK" in os.environ


def spawn_ranks(argv: Sequence[str], world: int, timeout_s: Optional[float] = None,
                extra_env: Optional[dict] = None) -> int:
    """Run ``python argv...`` as ``world`` ranks on this node; returns the exit code for the parent
    (0 only if every rank exited 0).  Rank 0's stdout is passed through; the other ranks' stdout is
    dropped (they print nothing by contract), every rank's stderr goes to ours."""
    if world < 1:
        raise ValueError("world must be positive")
    port = free_port()
    procs: List[subprocess.Popen] = []
    for rank in range(world):
        env = dict(os.environ)
        env.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_WORLD_SIZE": str(world),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this platform (RCCL needs it)
        if extra_env:
            env.update(extra_env)
        procs.append(subprocess.Popen([sys.executable, *argv], env=env,
                                      stdout=None if rank == 0 else subprocess.DEVNULL))
    deadline = None if timeout_s is None else time.monotonic() + timeout_s
    rc = 0
    live = list(procs)
    while live:
        for p in list(live):
            r = p.poll()
            if r is None:
                continue
            live.remove(p)
            if r != 0 and rc == 0:
                rc = r if r > 0 else 1
        if rc != 0 or (deadline is not None and time.monotonic() > deadline):
            if rc == 0:
                rc = 124
            for p in live:                      # exactly the processes started above
                p.terminate()
            for p in live:
                try:
                    p.wait(10)
                except subprocess.TimeoutExpired:
                    p.kill()
            break
        time.sleep(0.05)
    return rc
