"""globaltest drop-in on the GPU vs the reference's globaltest run on the CPU oracle net
(tests/golden/eval_metrics.json)."""
import numpy as np
import pytest

from tests.helpers import load_golden, make_args
from tests.test_local_training_gpu import SynthDataset

pytestmark = pytest.mark.gpu


def test_globaltest_matches_reference():
    from fedmlp_amd.model import build_model
    from fedmlp_amd.evaluations import globaltest
    g = load_golden("eval_metrics.json")
    args = make_args(n_classes=g["C"], batch_size=g["bs"], seed=g["init_seed"])
    ds = SynthDataset(g["N"], g["C"], g["hw"], g["data_seed"], False)
    res = globaltest(build_model(args), ds, args)
    for k, want in g["metrics"].items():
        assert abs(float(res[k]) - want) <= 1e-4 * abs(want) + 1e-6, (k, float(res[k]), want)
