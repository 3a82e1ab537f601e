"""The thin FL driver (main.py's FedMLP / FedAVG rows on the HIP engine) runs a short two-stage
schedule end to end on one GPU with device-resident models (ResidentNet) and finite losses."""
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run(argv):
    from fedmlp_amd import driver
    old = sys.argv
    sys.argv = ["driver"] + argv
    try:
        return driver.main()
    finally:
        sys.argv = old


def test_driver_fedmlp_two_stage_smoke():
    log = _run(["--exp", "FedMLP", "--n_clients", "2", "--n_classes", "4", "--rounds_warmup", "3",
                "--rounds_FedMLP_stage1", "2", "--batch_size", "32", "--n_local", "448", "--hw", "64", "--pretrained", "0"])
    assert len(log) == 3
    assert all(np.isfinite(r["mean_loss"]) for r in log)
    assert log[1]["mean_loss"] < log[0]["mean_loss"] * 1.5


def test_driver_two_stage_with_tail_batches():
    """--n_local 97 at --batch_size 32: every pass of every client ends with a batch of ONE sample (two views), the
    prototype pass (batches of 4 * 32) is a single short batch, the stage-2 feature pass ends with one image.  The
    reference keeps such tails (DataLoader drop_last=False, utils/local_training.py:47-48) and normalises by
    args.batch_size (:956-959)."""
    log = _run(["--exp", "FedMLP", "--n_clients", "2", "--n_classes", "4", "--rounds_warmup", "4",
                "--rounds_FedMLP_stage1", "2", "--batch_size", "32", "--n_local", "97", "--hw", "64", "--pretrained", "0",
                "--clean_threshold", "0.05", "--noise_threshold", "0.1"])
    assert len(log) == 4 and all(np.isfinite(r["mean_loss"]) for r in log)
    log2 = _run(["--exp", "FedAVG+FixMatch", "--n_clients", "2", "--n_classes", "4", "--rounds_warmup", "2",
                 "--batch_size", "32", "--n_local", "100", "--hw", "64", "--pretrained", "0"])
    assert len(log2) == 2 and all(np.isfinite(r["mean_loss"]) for r in log2)


def test_driver_fedavg_smoke():
    log = _run(["--exp", "FedAVG", "--n_clients", "2", "--n_classes", "4", "--rounds_warmup", "2",
                "--batch_size", "32", "--n_local", "96", "--hw", "64", "--pretrained", "0"])
    assert len(log) == 2 and all(np.isfinite(r["mean_loss"]) for r in log)


def test_rccl_allreduce_accepts_engine_state_memory():
    """bench.py --gpus N / the driver all-reduce the engine's state arena IN PLACE over RCCL.  The arena is
    hipMalloc'ed by the C library (not by torch's allocator) and reaches torch.distributed as a
    __cuda_array_interface__ view, so exercise that exact call with the nccl backend (one rank is all a 1-GPU
    box allows; the N > 1 arithmetic is covered by the world_size-2 gloo test on CPU)."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from fedmlp_amd import spec
    from fedmlp_amd.engine import Engine
    e = Engine("Resnet18", 5, 64, 64, 8)
    flat, cnt = spec.init_state("Resnet18", 5, 3)
    e.set_state(flat, cnt)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29531", rank=0, world_size=1,
                            device_id=torch.device("cuda:0"))
    try:
        e.state_scale(0.5)
        st = e.state_tensor()
        before = st.clone()
        dist.all_reduce(st, op=dist.ReduceOp.SUM)
        torch.cuda.synchronize()
        assert torch.equal(st, before)
        got, _ = e.get_state()
        np.testing.assert_allclose(got, flat * np.float32(0.5), rtol=0, atol=0)
    finally:
        dist.destroy_process_group()
        e.close()
