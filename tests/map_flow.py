"""The `mAP vs ref` half of BASELINE.json's metric as a paired multi-seed study.

One run = the TWO-STAGE FedMLP flow of main.py:135-237 (stage-1 rounds with the frozen round-start teacher, prototype
pass, then stage-2 rounds with cosine tagging / pseudo-labels, FedAvg / FedAvg_tao / FedAvg_proto after every round) on
a synthetic partial-label problem with signal, followed by the test mAP / AUROC (utils/evaluations.py:41-49) of the final
global model on a 32 768-sample test set.  4 clients, client i annotates class i of C = 4 (every class has exactly one
annotating client), 6 % of the observed labels flipped in train and test so that the metric's ceiling is below 1.

Both sides run the SAME seeds: seed s fixes the training data, the initial weights and every batch order, so the two
sides differ only in their arithmetic -- a paired comparison.  The oracle side (CPU restatement, pinned on the reference's
trajectories in tests/test_oracle_golden.py) runs in the build container and is committed as
tests/golden/map_study_oracle.json; the HIP side runs on the GPU box (tests/test_eval_gpu.py)."""
import copy

import numpy as np
import torch

C, N_CL, N_LOCAL, HW, BS = 4, 4, 384, 32, 32
S1, ROUNDS, LR = 5, 10, 3e-4
N_TEST, TEST_SEED = 32768, 600
SIGNAL, LABEL_NOISE = 0.45, 0.06
SEEDS = list(range(32))     # (16 in round 3; doubled in round 4 to halve the standard error of the paired difference)


def _patterns():
    return torch.randn((C, 3, HW, HW), generator=torch.Generator().manual_seed(999))


def make_split(n, seed, two_view):
    """images = noise + signal * sum_c y_c pattern_c; observed labels = clean labels with LABEL_NOISE flipped"""
    g = torch.Generator().manual_seed(seed)
    clean = (torch.rand((n, C), generator=g) < 0.3).float()
    x = 0.7 * torch.randn((n, 3, HW, HW), generator=g) + SIGNAL * torch.einsum("nc,cdhw->ndhw", clean, _patterns())
    flip = (torch.rand((n, C), generator=g) < LABEL_NOISE).float()
    y = (clean + flip - 2 * clean * flip).numpy().astype(np.float32)
    x2 = x + 0.1 * torch.randn((n, 3, HW, HW), generator=g) if two_view else None
    return x, x2, y


def problem(seed):
    x1, x2, y = make_split(N_CL * N_LOCAL, 1000 + seed, True)
    users = [list(range(i * N_LOCAL, (i + 1) * N_LOCAL)) for i in range(N_CL)]
    rs = np.random.RandomState(2000 + seed)
    orders = [[(rs.permutation(N_LOCAL).tolist(), rs.permutation(N_LOCAL).tolist()) for _ in range(N_CL)] for _ in range(ROUNDS)]
    return x1, x2, y, users, orders


def args_for(seed):
    from tests.helpers import make_args
    return make_args(n_classes=C, n_clients=N_CL, batch_size=BS, seed=3000 + seed, base_lr=LR, rounds_FedMLP_stage1=S1,
                     pretrained=0)


def class_lists_of(y):
    pos = [np.where(y[:, c] == 1)[0] for c in range(C)]
    return pos, [p.copy() for p in pos]


CLS_ACT = [[c] for c in range(C)]                                    # class c is annotated by client c
CLS_NEG = [[i for i in range(N_CL) if i != c] for c in range(C)]


def run_oracle(seed, test):
    """-> (mAP, AUROC) of the final global model, CPU oracle"""
    from oracle import steps_ref as R
    from tests.helpers import oracle_net
    from fedmlp_amd.evaluations import multilabel_metrics
    x1, x2, y, users, orders = problem(seed)
    args = args_for(seed)
    data = {"targets": y, "image_aug_1": x1, "image_aug_2": x2}
    _, neg = class_lists_of(y)
    glob = oracle_net(C, args.seed)
    cls = [R.RefClient(args, i, data, users[i], neg, [i]) for i in range(N_CL)]
    prototype, lens = None, [N_LOCAL] * N_CL
    for rnd in range(ROUNDS):
        w, taos, protos = [], [], []
        for i, cl in enumerate(cls):
            net = copy.deepcopy(glob)
            fo, to = orders[rnd][i]
            if rnd < S1:
                ret = cl.stage1(net, to, with_proto=(rnd == S1 - 1), negative_param=cl.negative)
            else:
                ret = cl.stage2(rnd, net, prototype, cl.negative, fo, to)
            w.append(copy.deepcopy(ret[0]))
            if len(ret) >= 5:
                taos.append(ret[3]); protos.append(ret[4])
        glob.load_state_dict(R.fedavg(w, lens))
        if rnd >= S1 - 1:
            prototype = R.fedavg_proto(protos, lens, CLS_ACT)
    glob.eval()
    zs = []
    with torch.no_grad():
        for i in range(0, len(test[0]), 1024):
            zs.append(glob(test[0][i:i + 1024])[1])
    m = multilabel_metrics(test[2], torch.sigmoid(torch.cat(zs)).numpy())
    return float(m["mAP"]), float(m["auc"])


class _DS:
    """dataset/all_dataset.py:64-83 contract over in-memory tensors (+ HBM-resident views for the engine)"""

    def __init__(self, x1, x2, y):
        self.x1, self.x2, self.targets, self._v = x1, x2, y, None

    def __len__(self):
        return len(self.targets)

    def __getitem__(self, i):
        d = {"image": self.x1[i], "image_aug_1": self.x1[i], "target": self.targets[i].copy(), "index": i}
        if self.x2 is not None:
            d["image_aug_2"] = self.x2[i]
        return d

    def device_views(self, device):
        if self._v is None:
            self._v = {"image": self.x1.to(device), "image_aug_1": self.x1.to(device)}
            if self.x2 is not None:
                self._v["image_aug_2"] = self.x2.to(device)
        return self._v


def run_hip(seed, test_ds):
    """-> (mAP, AUROC) of the final global model, HIP engine through the drop-in surface"""
    from fedmlp_amd.model import build_model
    from fedmlp_amd.fedavg import FedAvg, FedAvg_tao, FedAvg_proto
    from fedmlp_amd.evaluations import globaltest
    from tests.helpers import replay_local_update
    LocalUpdate = replay_local_update()
    x1, x2, y, users, orders = problem(seed)
    args = args_for(seed)
    ds = _DS(x1, x2, y)
    pos, neg = class_lists_of(y)
    netglob = build_model(args)
    locs = [LocalUpdate(args, i, ds, users[i], pos, neg, active_class_list=[i]) for i in range(N_CL)]
    tao, Prototype, lens = [0] * C, [], [N_LOCAL] * N_CL
    for rnd in range(ROUNDS):
        w, taos, protos = [], [], []
        for i, loc in enumerate(locs):
            fo, to = orders[rnd][i]
            if rnd < S1:
                loc.order_queue.append(to)
                a1 = (None, None) if rnd < S1 - 1 else (loc.negative_class_list, loc.active_class_list)
            else:
                loc.order_queue += [fo, to]
                a1 = (loc.negative_class_list, loc.active_class_list)
            ret = loc.train_FedMLP(rnd, tao, Prototype, None, a1[0], a1[1], net=copy.deepcopy(netglob))
            w.append(copy.deepcopy(ret[0]))
            if len(ret) == 8:
                taos.append(ret[6]); protos.append(ret[7])
        netglob.load_state_dict(FedAvg(w, lens))
        if rnd >= S1 - 1:
            tao = FedAvg_tao(taos, lens, CLS_NEG)
            Prototype = FedAvg_proto(protos, lens, CLS_ACT)
    m = globaltest(netglob, test_ds, args)
    return float(m["mAP"]), float(m["auc"])


def summarise(vals):
    a = np.asarray(vals, np.float64)
    return {"mean": float(a.mean()), "se": float(a.std(ddof=1) / np.sqrt(len(a))) if len(a) > 1 else None, "n": len(a)}
