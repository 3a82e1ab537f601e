"""The library's own RCCL communicator with a REAL world of 2 (fm_comm_init, fm_fedavg_allreduce / _tao / _proto) against the
torch.distributed fallback of fedmlp_amd/fedavg.py on the same inputs.  Needs two GPUs: skipped on the 1-GPU test box, runs
wherever `torch.cuda.device_count() >= 2` (the driver's 8-GPU node)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, json
import numpy as np, torch
import torch.distributed as dist
sys.path.insert(0, os.environ["FM_ROOT"])
from fedmlp_amd import spec
from fedmlp_amd.engine import Engine
from fedmlp_amd.fedavg import comm_init, fedavg_allreduce, tao_allreduce, proto_allreduce
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(rank)
dev = torch.device("cuda", rank)
dist.init_process_group("nccl", device_id=dev)
C = 5
res = {}
for use_lib in (True, False):
    e = Engine("Resnet18", C, 64, 64, 8, device=str(dev))
    flat, cnt = spec.init_state("Resnet18", C, 100 + rank)
    e.set_state(flat, cnt + 3 * rank)
    if use_lib:
        assert comm_init(e) == world
    w = [0.3, 0.7][rank]
    st = fedavg_allreduce(e, w)
    got, gcnt = e.get_state()
    rs = np.random.RandomState(7 + rank)
    t = rs.uniform(size=C); proto = rs.standard_normal((2 * C, 512)).astype(np.float32)
    neg = [0.0 if c == rank else 1.0 for c in range(C)]; act = [1.0 if c == rank else 0.0 for c in range(C)]
    tao = tao_allreduce(t, [300, 700][rank], neg, device=dev, engine=e if use_lib else None)
    pr = proto_allreduce(proto, [300, 700][rank], act, device=dev, engine=e if use_lib else None)
    res["lib" if use_lib else "dist"] = (got, gcnt, np.asarray(tao), pr.numpy())
    e.close()
if rank == 0:
    for k in range(4):
        a, b = res["lib"][k], res["dist"][k]
        assert np.array_equal(np.isnan(a), np.isnan(b)), k
        np.testing.assert_allclose(a, b, rtol=2e-6, atol=1e-7, equal_nan=True, err_msg=str(k))
    print("RCCL_WORLD2_OK", flush=True)
dist.barrier(); dist.destroy_process_group()
'''


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (the library's RCCL communicator with a world of 2)")
def test_library_rccl_world2_matches_torch_distributed(tmp_path):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, FM_ROOT=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), str(script)], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "RCCL_WORLD2_OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
