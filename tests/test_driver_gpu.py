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
                "--rounds_FedMLP_stage1", "2", "--batch_size", "32", "--n_local", "448", "--hw", "64"])
    assert len(log) == 3
    assert all(np.isfinite(r["mean_loss"]) for r in log)
    assert log[1]["mean_loss"] < log[0]["mean_loss"] * 1.5


def test_driver_fedavg_smoke():
    log = _run(["--exp", "FedAVG", "--n_clients", "2", "--n_classes", "4", "--rounds_warmup", "2",
                "--batch_size", "32", "--n_local", "96", "--hw", "64"])
    assert len(log) == 2 and all(np.isfinite(r["mean_loss"]) for r in log)
