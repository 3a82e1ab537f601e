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


class _Learnable:
    """Synthetic data with signal: image = noise + sum_c y_c * pattern_c, so mAP after training means something."""

    def __init__(self, n, C, hw, seed):
        import torch
        g = torch.Generator().manual_seed(seed)
        self.patterns = torch.randn((C, 3, hw, hw), generator=torch.Generator().manual_seed(999))
        self.targets = (torch.rand((n, C), generator=g) < 0.3).float().numpy()
        y = torch.from_numpy(self.targets)
        self.x = 0.7 * torch.randn((n, 3, hw, hw), generator=g) + 0.6 * torch.einsum("nc,cdhw->ndhw", y, self.patterns)
        self._v = None

    def __len__(self):
        return len(self.targets)

    def __getitem__(self, i):
        return {"image": self.x[i], "target": self.targets[i].copy(), "index": i}

    def device_views(self, device):
        if self._v is None:
            self._v = {"image": self.x.to(device)}
        return self._v


def test_map_after_training_matches_oracle():
    """The 'mAP vs ref' half of the metric on data with signal: 2 clients x 4 FedAvg rounds of LocalUpdate.train on
    the GPU and the same flow through the CPU oracle (same init, same batch orders) end at the same test mAP
    (~0.94 against a prevalence of 0.3).  On this small problem (192 training / 1024 test samples, 24 Adam steps from
    random init) the CPU oracle's OWN mAP spreads over 0.9354 ... 0.9403 between 1, 4 and 8 threads and under a
    +-1e-6 weight perturbation (and 0.923 ... 0.928 on a 256-sample test set across hosts), so the test allows
    1.5 % absolute; the +-0.2 % of the north star is a statement about converged training on a real test set."""
    import copy
    import torch
    from fedmlp_amd.model import build_model
    from fedmlp_amd.local_training import LocalUpdate
    from fedmlp_amd.fedavg import FedAvg
    from fedmlp_amd.evaluations import globaltest, multilabel_metrics
    from oracle import steps_ref as R
    from tests.helpers import oracle_net
    from tests.synth import class_lists
    C, N, hw, n_cl, rounds = 4, 96, 32, 2, 4
    args = make_args(n_classes=C, n_clients=n_cl, batch_size=32, seed=21, base_lr=3e-4)
    train, test = _Learnable(n_cl * N, C, hw, 5), _Learnable(1024, C, hw, 6)
    pos, neg = class_lists(train.targets, C)
    users = [list(range(i * N, (i + 1) * N)) for i in range(n_cl)]
    rs = np.random.RandomState(3)
    orders = [[rs.permutation(N).tolist() for _ in range(n_cl)] for _ in range(rounds)]
    # ---- GPU product
    netglob = build_model(args)
    locs = [LocalUpdate(args, i, train, users[i], pos, neg, active_class_list=list(range(C))) for i in range(n_cl)]
    for r in range(rounds):
        w = []
        for i in range(n_cl):
            locs[i].order_queue.append(orders[r][i])
            w.append(copy.deepcopy(locs[i].train(r, copy.deepcopy(netglob), None)[0]))
        netglob.load_state_dict(FedAvg(w, [N] * n_cl))
    got = globaltest(netglob, test, args)
    # ---- CPU oracle
    data = {"targets": train.targets, "image": train.x}
    glob = oracle_net(C, 21)
    cls = [R.RefClient(args, i, data, users[i], neg, list(range(C))) for i in range(n_cl)]
    for r in range(rounds):
        w = [copy.deepcopy(cls[i].train(copy.deepcopy(glob), orders[r][i])[0]) for i in range(n_cl)]
        glob.load_state_dict(R.fedavg(w, [N] * n_cl))
    glob.eval()
    with torch.no_grad():
        _, z = glob(test.x)
    want = multilabel_metrics(test.targets, torch.sigmoid(z).numpy())
    g_map, w_map = float(got["mAP"]), float(want["mAP"])
    base = float((test.targets.mean(0)).mean())          # mAP of a random scorer ~ prevalence
    import json, os
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/parity_map.json", "w") as f:
        json.dump({"mAP_hip": g_map, "mAP_oracle": w_map, "auc_hip": float(got["auc"]), "auc_oracle": float(want["auc"]),
                   "prevalence": base}, f, indent=1)
    assert w_map > base + 0.4, ("the oracle did not learn", w_map, base)
    assert abs(g_map - w_map) < 1.5e-2, (g_map, w_map)
    assert abs(float(got["auc"]) - float(want["auc"])) < 1.5e-2
