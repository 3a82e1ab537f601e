"""One process per GPU: start N ranks of a script on this node.

The reference trains its clients one after another in ONE process on ONE GPU
(main.py:32, 135-184).  Here a round's clients run side by side, one rank per
GPU, and only FedAvg (utils/FedAvg.py:7-14) crosses ranks (RCCL all-reduce).

`python bench.py --gpus N` / `python -m fedmlp_amd.driver --gpus N` call
spawn_ranks() BEFORE anything touches the GPU: the parent only waits for
`python -m torch.distributed.run` (a child process) and exits with its code.
A process that has initialised HIP is never re-exec'ed.
"""
import os
import socket
import subprocess
import sys


def default_device():
    """cuda:<LOCAL_RANK> for a rank started by torch.distributed.run, else the current device."""
    import torch
    if "LOCAL_RANK" in os.environ:
        return f"cuda:{int(os.environ['LOCAL_RANK'])}"
    return f"cuda:{torch.cuda.current_device()}" if torch.cuda.is_available() else "cuda:0"


def launched_by_torchrun():
    return "WORLD_SIZE" in os.environ and "RANK" in os.environ


def free_port():
    s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def rank_command(script, argv, nproc, port=None, module=False):
    """The command line the driver itself uses for N > 1 (prompt contract)."""
    port = port or free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
           "--master-addr", "127.0.0.1", "--master-port", str(port)]
    cmd += (["-m", script] if module else [script])
    return cmd + list(argv)


def spawn_ranks(script, argv, nproc, module=False, env=None, timeout=None):
    """Run `script argv...` as nproc ranks (children of this process); returns the exit code.
    stdout/stderr are inherited, so rank 0's JSON line is this process's output too."""
    e = dict(os.environ)
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC (RCCL across processes on this pool)
    e.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // max(nproc, 1))))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    if env:
        e.update(env)
    p = subprocess.Popen(rank_command(script, argv, nproc, module=module), env=e)
    try:
        return p.wait(timeout=timeout)
    except subprocess.TimeoutExpired:
        p.kill()              # the exact child we started, never a pattern
        p.wait()
        return 124
