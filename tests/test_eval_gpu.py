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

    def __init__(self, n, C, hw, seed, signal=0.6, label_noise=0.0):
        import torch
        g = torch.Generator().manual_seed(seed)
        self.patterns = torch.randn((C, 3, hw, hw), generator=torch.Generator().manual_seed(999))
        clean = (torch.rand((n, C), generator=g) < 0.3).float()
        self.x = 0.7 * torch.randn((n, 3, hw, hw), generator=g) + signal * torch.einsum("nc,cdhw->ndhw", clean, self.patterns)
        flip = (torch.rand((n, C), generator=g) < label_noise).float() if label_noise else torch.zeros_like(clean)
        self.targets = (clean + flip - 2 * clean * flip).numpy()        # observed labels (a fraction flipped: mAP < 1)
        self._v = None

    def __len__(self):
        return len(self.targets)

    def __getitem__(self, i):
        return {"image": self.x[i], "target": self.targets[i].copy(), "index": i}

    def device_views(self, device):
        if self._v is None:
            self._v = {"image": self.x.to(device)}
        return self._v


CONVERGED = dict(N=512, rounds=10, lr=3e-4, ntest=4096, signal=0.25, label_noise=0.06, avg_last=3)
CONVERGED_ORDERS = (3, 4, 5)


def _map_problem(N, rounds, lr, ntest, signal, label_noise, order_seed):
    from tests.synth import class_lists
    C, hw, n_cl = 4, 32, 2
    args = make_args(n_classes=C, n_clients=n_cl, batch_size=32, seed=21, base_lr=lr)
    train = _Learnable(n_cl * N, C, hw, 5, signal, label_noise)
    test = _Learnable(ntest, C, hw, 6, signal, label_noise)
    pos, neg = class_lists(train.targets, C)
    users = [list(range(i * N, (i + 1) * N)) for i in range(n_cl)]
    rs = np.random.RandomState(order_seed)
    orders = [[rs.permutation(N).tolist() for _ in range(n_cl)] for _ in range(rounds)]
    return C, n_cl, args, train, test, pos, neg, users, orders


def map_flow_oracle(N, rounds, lr, ntest, signal=0.6, label_noise=0.0, avg_last=1, order_seed=3):
    """CPU-oracle half: [(mAP, AUROC)] of the last `avg_last` rounds (also run by tools/map_study.py converged)."""
    import copy
    import torch
    from fedmlp_amd.evaluations import multilabel_metrics
    from oracle import steps_ref as R
    from tests.helpers import oracle_net
    C, n_cl, args, train, test, pos, neg, users, orders = _map_problem(N, rounds, lr, ntest, signal, label_noise, order_seed)
    data = {"targets": train.targets, "image": train.x}
    glob = oracle_net(C, 21)
    cls = [R.RefClient(args, i, data, users[i], neg, list(range(C))) for i in range(n_cl)]
    want = []
    for r in range(rounds):
        w = [copy.deepcopy(cls[i].train(copy.deepcopy(glob), orders[r][i])[0]) for i in range(n_cl)]
        glob.load_state_dict(R.fedavg(w, [N] * n_cl))
        if r >= rounds - avg_last:
            glob.eval()
            with torch.no_grad():
                _, z = glob(test.x)
            m = multilabel_metrics(test.targets, torch.sigmoid(z).numpy())
            want.append((float(m["mAP"]), float(m["auc"])))
    return want


def _map_flow(N, rounds, lr, ntest, tol, report, signal=0.6, label_noise=0.0, avg_last=1, order_seed=3, oracle=None):
    """2 clients x `rounds` FedAvg rounds of LocalUpdate.train on the GPU and through the CPU oracle (same init, same
    batch orders; `oracle` = its recorded per-round metrics when it was run elsewhere); test mAP / AUROC averaged over the
    last `avg_last` rounds on both sides."""
    import copy
    import json
    import os
    from fedmlp_amd.model import build_model
    from tests.helpers import replay_local_update
    LocalUpdate = replay_local_update()      # LocalUpdate + recorded batch orders / tagging log (tests/helpers.py)
    from fedmlp_amd.fedavg import FedAvg
    from fedmlp_amd.evaluations import globaltest
    C, n_cl, args, train, test, pos, neg, users, orders = _map_problem(N, rounds, lr, ntest, signal, label_noise, order_seed)
    # ---- GPU product
    netglob = build_model(args)
    locs = [LocalUpdate(args, i, train, users[i], pos, neg, active_class_list=list(range(C))) for i in range(n_cl)]
    got = []
    for r in range(rounds):
        w = []
        for i in range(n_cl):
            locs[i].order_queue.append(orders[r][i])
            w.append(copy.deepcopy(locs[i].train(r, copy.deepcopy(netglob), None)[0]))
        netglob.load_state_dict(FedAvg(w, [N] * n_cl))
        if r >= rounds - avg_last:
            m = globaltest(netglob, test, args)
            got.append((float(m["mAP"]), float(m["auc"])))
    # ---- CPU oracle
    want = oracle if oracle is not None else map_flow_oracle(N, rounds, lr, ntest, signal, label_noise, avg_last, order_seed)
    g_map, g_auc = np.mean([v[0] for v in got]), np.mean([v[1] for v in got])
    w_map, w_auc = np.mean([v[0] for v in want]), np.mean([v[1] for v in want])
    base = float((test.targets.mean(0)).mean())          # mAP of a random scorer ~ prevalence
    os.makedirs("gpurun_out", exist_ok=True)
    with open(f"gpurun_out/{report}", "w") as f:
        json.dump({"mAP_hip": float(g_map), "mAP_oracle": float(w_map), "auc_hip": float(g_auc), "auc_oracle": float(w_auc),
                   "per_round_hip": got, "per_round_oracle": want, "prevalence": base}, f, indent=1)
    assert w_map > base + 0.4, ("the oracle did not learn", w_map, base)
    assert abs(g_map - w_map) < tol, (g_map, w_map)
    assert abs(g_auc - w_auc) < tol
    return float(g_map - w_map), float(g_auc - w_auc)


def test_map_after_training_matches_oracle():
    """The 'mAP vs ref' half of the metric on data with signal: 2 clients x 4 FedAvg rounds of LocalUpdate.train on
    the GPU and the same flow through the CPU oracle (same init, same batch orders) end at the same test mAP
    (~0.94 against a prevalence of 0.3).  On this small problem (192 training / 1024 test samples, 24 Adam steps from
    random init) the CPU oracle's OWN mAP spreads over 0.9354 ... 0.9403 between 1, 4 and 8 threads and under a
    +-1e-6 weight perturbation (and 0.923 ... 0.928 on a 256-sample test set across hosts), so the test allows
    1.5 % absolute; the +-0.2 % of the north star is a statement about converged training on a real test set."""
    _map_flow(96, 4, 3e-4, 1024, 1.5e-2, "parity_map_small.json")


def test_map_after_converged_training_matches_oracle():
    """The same flow trained to convergence on a problem whose ceiling is below 1 (weak patterns, 6 % of the labels
    flipped in train and test; 2 clients x 512 samples, 10 FedAvg rounds = 160 Adam steps per client, 4096 test
    samples), for THREE batch orders; the oracle half was run in the build container (`python tools/map_study.py
    converged` -> tests/golden/map_converged_oracle.json; 2.3 minutes per order on 8 cores).  Once the fit has converged the
    chaotic early trajectory no longer decides the metric, and the GPU and CPU-oracle models land on the same plateau --
    but a single pair is still one draw of a chaotic optimisation: the oracle alone moves by +-0.4 % from round to round
    on that plateau and by 0.3-0.5 % between two hosts (order 3: 0.8342 in the build container, 0.8325 and 0.8364 on two GPU
    boxes' hosts), and the HIP side of that pair measured 0.8302, 0.8284 and 0.8216 under three roundings of the engine's
    GEMMs (fp32 MFMA of rounds 3 and 4, the bf16-partial-product form).  So one pair is bounded by 2 % absolute and the
    statement is the MEAN over the three orders, below 0.75 % (measured under round 5's kernels, profiles/r05/parity_map_converged.json: mAP -0.49 / +0.50 / +0.52 %, mean
    +0.18 %; AUROC +0.03 %; round 4's kernels: -0.14 / -0.13 / +0.34 %).  The 32-seed paired study
    below is the statistical form; the north star's +-0.2 % needs a real dataset to be decidable."""
    import json
    g = load_golden("map_converged_oracle.json")
    assert g["config"] == {k: CONVERGED[k] for k in g["config"]}
    d = [_map_flow(tol=2e-2, report=f"parity_map_converged_order{o}.json", order_seed=o,
                   oracle=[tuple(v) for v in g["orders"][str(o)]], **CONVERGED) for o in CONVERGED_ORDERS]
    with open("gpurun_out/parity_map_converged.json", "w") as f:
        json.dump({"orders": list(CONVERGED_ORDERS), "mAP_hip_minus_oracle": [v[0] for v in d],
                   "auc_hip_minus_oracle": [v[1] for v in d]}, f)
    assert abs(np.mean([v[0] for v in d])) < 7.5e-3, d
    assert abs(np.mean([v[1] for v in d])) < 7.5e-3, d


def test_map_two_stage_flow_paired_study():
    """BASELINE.json's `mAP vs ref` as a measurable statement: the TWO-STAGE FedMLP flow (tests/map_flow.py: 4 clients, 5
    stage-1 + 5 stage-2 rounds, 6 % label noise) for the same 16 seeds on both sides, test mAP / AUROC of the final global
    model on a 32 768-sample test set.  The oracle half was run in the build container (tools/map_study.py ->
    tests/golden/map_study_oracle.json); this is the HIP half.  Paired differences d_s = HIP_s - oracle_s:
    |mean d| <= 0.2 % (the north star's figure), or <= 2 standard errors of the mean with s.e. <= 0.1 % -- a single pair
    differs by a few tenths of a percent either way (two trajectories of a chaotic optimisation), the MEANS agree."""
    import json
    import os
    from tests import map_flow as F
    from tests.helpers import load_golden
    g = load_golden("map_study_oracle.json")
    seeds = sorted(int(s) for s in g["runs"])
    assert len(seeds) >= 8, "the committed oracle fixture must hold at least 8 seeds"
    x, _, y = F.make_split(F.N_TEST, F.TEST_SEED, False)
    test_ds = F._DS(x, None, y)
    runs, d_map, d_auc = {}, [], []
    for s in seeds:
        m, a = F.run_hip(s, test_ds)
        runs[str(s)] = {"mAP": m, "auc": a}
        d_map.append(m - g["runs"][str(s)]["mAP"]); d_auc.append(a - g["runs"][str(s)]["auc"])
    rep = {"config": g["config"], "prevalence": g["prevalence"], "seeds": seeds,
           "hip": {"runs": runs, "mAP": F.summarise([r["mAP"] for r in runs.values()]), "auc": F.summarise([r["auc"] for r in runs.values()])},
           "oracle": {"runs": g["runs"], "mAP": g["mAP"], "auc": g["auc"]},
           "paired_difference_hip_minus_oracle": {"mAP": F.summarise(d_map), "auc": F.summarise(d_auc),
                                                  "max_abs_mAP": float(np.abs(d_map).max())}}
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/parity_map.json", "w") as f:
        json.dump(rep, f, indent=1)
    assert g["mAP"]["mean"] > g["prevalence"] + 0.4, "the oracle did not learn"
    for k in ("mAP", "auc"):
        d = rep["paired_difference_hip_minus_oracle"][k]
        assert abs(d["mean"]) <= 2e-3 or (abs(d["mean"]) <= 2 * d["se"] and d["se"] <= 1e-3), (k, d)
