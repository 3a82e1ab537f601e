"""Round-2 goldens (made by tests/golden/make_golden.py importing /root/reference in the build container):
  * step_full.json     -- ONE FedMLP stage-1 step through the reference's own train_FedMLP at the BENCHMARKED
                          size (bs 128, two 3x224x224 views, C = 5): loss and every parameter gradient
  * traj_fedmlp64.*    -- the two-stage FedMLP flow on a conditioned problem (64x64, 2 x 1024 samples, non-trivial BN
                          affine): fixed tolerances (norms 1e-3, prototypes / probe logits 1e-2), no oracle-sensitivity
                          excuses
  * traj_fedmlp_c14.*  -- the same flow with 14 labels and 3 clients (BASELINE configs[2] shape): NaN prototype rows
                          for the 11 classes nobody annotates, tagging only where a prototype exists
"""
import copy
import json
import os

import numpy as np
import pytest
import torch

from fedmlp_amd import spec
from tests.helpers import load_golden, make_args, GOLDEN
from tests.synth import synth_arrays, class_lists, perturbed_bn
from tests.test_local_training_gpu import SynthDataset, _norms

pytestmark = pytest.mark.gpu


def _dump(report, name):
    os.makedirs("gpurun_out", exist_ok=True)
    with open(os.path.join("gpurun_out", name), "w") as f:
        json.dump(report, f, indent=1)


def _init_net(args, bn_seed):
    """build_model(args) + the golden's BatchNorm perturbation (tests/synth.perturbed_bn)"""
    from fedmlp_amd.model import build_model
    net = build_model(args)
    if bn_seed is not None:
        sd = net.state_dict()
        for k, v in perturbed_bn([(k, tuple(t.shape)) for k, t in sd.items()], bn_seed):
            sd[k] = torch.from_numpy(v)
        net.load_state_dict(sd)
    return net


def test_stage1_step_at_the_benchmarked_size_matches_the_reference():
    """bs 128 x 2 views x 224x224: the configuration bench.py times (512 persistent igemm blocks, multi-segment
    stream-K fix-ups, 1024-block wgrad splits, pixel-tile-major tile order) against the reference's own step."""
    from fedmlp_amd.engine import Engine
    g = load_golden("step_full.json")
    C, N, hw = g["C"], g["N"], g["hw"]
    targets, x1, x2 = synth_arrays(N, C, hw, g["data_seed"], True)
    args = make_args(n_classes=C, n_clients=1, batch_size=g["bs"], seed=g["init_seed"], pretrained=0)
    net = _init_net(args, g["bn_seed"])
    flat, cnt = spec.state_dict_to_flat("Resnet18", C, net.state_dict())
    eng = Engine("Resnet18", C, hw, hw, 2 * N)
    try:
        eng.set_state(flat, cnt)
        eng.teacher_snapshot()
        eng.adam_reset(args.base_lr)
        pos, _ = class_lists(targets, C)
        y = targets.copy()
        y[:, 1:] = 0.0                                   # client 0 annotates class 0 only (DatasetSplit masking)
        lo = torch.zeros(1, device=eng.device)
        eng.step_stage1(torch.from_numpy(x1).to(eng.device), torch.from_numpy(x2).to(eng.device),
                        torch.from_numpy(y).to(eng.device), [1.0, 0, 0, 0, 0], 1, g["bs"], lo)
        report = {"loss": lo.item(), "loss_ref": g["loss"], "loss_rel_err": abs(lo.item() - g["loss"]) / abs(g["loss"])}
        gsd = spec.flat_to_state_dict("Resnet18", C, eng.debug_get_grads(), np.zeros(eng.ni, np.int64))
        typ = float(np.median([v["absmax"] for v in g["grads"].values()]))
        worst = {}
        for k, w in g["grads"].items():
            got = gsd[k].astype(np.float64)
            e_norm = abs(np.linalg.norm(got) - w["norm"]) / (w["norm"] + 1e-3 * typ)
            e_sum = abs(got.sum() - w["sum"]) / (w["norm"] * np.sqrt(got.size) + 1e-30)
            e_head = np.abs(got.ravel()[:3] - np.array(w["head"])).max() / (w["absmax"] + 1e-30)
            worst[k] = max(e_norm, e_sum, e_head)
            for nm, v in (("norm", e_norm), ("sum", e_sum), ("head", e_head)):
                report["max_" + nm + "_err"] = max(report.get("max_" + nm + "_err", 0.0), float(v))
        k_bad = max(worst, key=worst.get)
        report["worst_grad_tensor"] = k_bad
        report["worst_grad_err"] = worst[k_bad]
        report["median_grad_err"] = float(np.median(list(worst.values())))
        _dump(report, "parity_step_full.json")
        assert report["loss_rel_err"] < 1e-5, report
        # Against a STORED reference result the engine's ReLU masks cannot be handed to the other side (the 64x64 step
        # tests do that and hold 2e-4): with 6.4e8 ReLU inputs in this step a few hundred lie within rounding distance
        # of zero.  Measured: worst tensor 3.0e-3 (a BN bias of layer 1), median 1.1e-3.
        assert worst[k_bad] < 6e-3 and report["median_grad_err"] < 2.5e-3, (k_bad, worst[k_bad])
        flat2, _ = eng.get_state()
        sd2 = spec.flat_to_state_dict("Resnet18", C, flat2, np.zeros(eng.ni, np.int64))
        for k, w in g["norms"].items():
            if "num_batches" in k:
                continue
            gn = float(np.linalg.norm(sd2[k].astype(np.float64)))
            assert abs(gn - w) <= 1e-4 * abs(w) + 1e-6, (k, gn, w)
    finally:
        eng.close()


def _replay_two_stage(name, tol, replace_near_ties=True):
    """The FedMLP two-stage flow (stage 1, prototype pass, tagging + selection, stage 2, FedAvg*) against a golden
    trajectory of the reference.  tol: fixed bounds {loss, norm, bn_bias_norm, proto, logits, t_count}.
    replace_near_ties=False leaves the engine's own picks in place (free-running): near-ties are then checked in the
    first stage-2 round only, because afterwards the two runs' pools differ."""
    from fedmlp_amd.local_training import LocalUpdate
    from fedmlp_amd.fedavg import FedAvg, FedAvg_tao, FedAvg_proto
    g = load_golden(name + ".json")
    P = np.load(os.path.join(GOLDEN, name + "_protos.npz"))
    C, n_cl, N, S1 = g["C"], g["n_clients"], g["N"], g["S1"]
    args = make_args(n_classes=C, n_clients=n_cl, rounds_FedMLP_stage1=S1, seed=g["init_seed"], pretrained=0)
    ds = SynthDataset(n_cl * N, C, g["hw"], g["data_seed"], True, g.get("p_pos", 0.3))
    pos, neg = class_lists(ds.targets, C)
    netglob = _init_net(args, g.get("bn_seed"))
    locs = [LocalUpdate(args, i, ds, g["users"][i], pos, neg, active_class_list=[i]) for i in range(n_cl)]
    for l in locs:
        l.tagging_log = []
    cur = {}                                   # round record of the golden being replayed

    def make_hook(i):
        """Discrete picks against the reference's.  The index work itself is pinned bit-exactly on identical inputs by
        tests/test_kat_gpu.py; here the inputs are features of a net trained on another machine, so a pick may differ
        ONLY where the engine's own similarities of the two candidates are a near-tie: <= 5e-3 of the row's range in
        the first stage-2 round, <= 2e-2 in later ones (their features come from a net that has meanwhile trained a
        round on pseudo-labels; its probe logits are 2-3e-2 of their range from the reference's by then) -- checked
        for every differing element of every stage-2 round.  A near-tie pick that differs is then REPLACED by
        the reference's: otherwise the two runs train on different pseudo-labelled sets from that round on and the
        continuous comparisons below (loss, norms, prototypes, t) would compare different experiments."""
        def hook(rnd, k, cls, clean, noise):
            r, prev = cur["r"], cur["prev"]
            out = []
            for side, got in ((0, clean), (1, noise)):
                full = r["traindata_idx"][i][2 * k + side]
                want = full[len(prev["traindata_idx"][i][2 * k + side]):] if prev is not None and rnd > S1 else full
                assert len(got) == len(want), (rnd, i, cls, side, len(got), len(want))
                got_s, want_s = set(got), set(want)
                if not replace_near_ties and rnd > S1:
                    out.append(list(got))
                    continue
                if got_s != want_s:
                    t = [t for t in locs[i].tagging_log if t["rnd"] == rnd and t["cls"] == cls][-1]
                    where = {v: j for j, v in enumerate(t["pool_idx"])}
                    rng = float(np.nanmax(t["sim"]) - np.nanmin(t["sim"]))
                    for gi, wi in zip(sorted(got_s - want_s), sorted(want_s - got_s)):
                        gap = abs(float(t["sim"][where[gi]]) - float(t["sim"][where[wi]]))
                        mx["pick_gap"] = max(mx["pick_gap"], gap / rng)
                        assert gap <= (5e-3 if rnd == S1 else 2e-2) * rng, (rnd, i, cls, gi, wi, gap, rng)
                    rep["picks_replaced"] += len(got_s - want_s)
                rep["picks_total"] += len(want)
                out.append(list(want) if replace_near_ties else list(got))
            return out[0], out[1]
        return hook

    for i, l in enumerate(locs):
        l.selection_hook = make_hook(i)
    tao, Prototype = [0] * C, []
    neg_lists, act_lists = g["neg_lists"], g["act_lists"]
    rep = {"max": {"loss": 0.0, "norm": 0.0, "bn_bias_norm": 0.0, "proto": 0.0, "logits": 0.0, "t_count": 0.0,
                   "pick_gap": 0.0},
           "picks_replaced": 0, "picks_total": 0}
    mx = rep["max"]

    def cmp_norms(got, want):
        for k, w in want.items():
            if "num_batches" in k:
                assert abs(got[k] - w) < 0.5, k
                continue
            kind = "bn_bias_norm" if (k.endswith(".bias") and not k.startswith("fc.")) else "norm"
            mx[kind] = max(mx[kind], abs(got[k] - w) / (abs(w) + 1e-12))

    for rnd, r in enumerate(g["rounds"]):
        cur["r"], cur["prev"] = r, (g["rounds"][rnd - 1] if rnd > 0 else None)
        w, taos, protos = [], [], []
        for i in range(n_cl):
            if rnd < S1:
                locs[i].order_queue.append(r["train_orders"][i])
                a1 = (None, None) if rnd < S1 - 1 else (neg_lists[i], act_lists[i])
            else:
                locs[i].order_queue += [r["feat_orders"][i], r["train_orders"][i]]
                a1 = (neg_lists[i], act_lists[i])
            ret = locs[i].train_FedMLP(rnd, tao, Prototype, None, a1[0], a1[1], net=copy.deepcopy(netglob))
            mx["loss"] = max(mx["loss"], abs(ret[1] - r["loss"][i]) / abs(r["loss"][i]))
            cmp_norms(_norms(ret[0]), r["norms"][i])
            if rnd == 0:
                assert ret[4] == neg_lists[i] and ret[5] == act_lists[i]
            if rnd >= S1:
                # after the hook the lists equal the reference's as sets (membership is all DatasetSplit_pseudo uses,
                # :1462-1469); how many picks the hook had to replace is in the report
                if replace_near_ties:
                    assert [sorted(a) for a in locs[i].traindata_idx] == [sorted(b) for b in r["traindata_idx"][i]]
            w.append(copy.deepcopy(ret[0]))
            if len(ret) == 8:
                taos.append(ret[6]); protos.append(ret[7])
                mx["t_count"] = max(mx["t_count"], float(np.abs(ret[6] - P[f"r{rnd}_c{i}_t"]).max() * N))
                want = P[f"r{rnd}_c{i}_proto"]
                assert np.array_equal(np.isnan(ret[7].numpy()), np.isnan(want))
                mx["proto"] = max(mx["proto"], float(np.nanmax(np.abs(ret[7].numpy() - want)) / np.nanmax(np.abs(want))))
        netglob.load_state_dict(FedAvg(w, [N] * n_cl))
        if rnd >= S1 - 1:
            tao = FedAvg_tao(taos, [N] * n_cl, g["class_negative_client_list"])
            Prototype = FedAvg_proto(protos, [N] * n_cl, g["class_active_client_list"])
            want = P[f"r{rnd}_glob_proto"]
            assert np.array_equal(np.isnan(Prototype.numpy()), np.isnan(want))              # NaN rows
            mx["proto"] = max(mx["proto"], float(np.nanmax(np.abs(Prototype.numpy() - want)) / np.nanmax(np.abs(want))))
            mx["t_count"] = max(mx["t_count"], float(np.abs(tao - np.array(r["tao"])).max() * N))
        cmp_norms(_norms(netglob.state_dict()), r["glob_norms"])
        netglob.eval()
        _, z = netglob(ds.x1[:4])
        want = np.array(r["probe_logits"])
        mx["logits"] = max(mx["logits"], float(np.abs(z.cpu().numpy() - want).max() / np.abs(want).max()))
        rep.setdefault("running_max_after_round", []).append(dict(mx))
    _dump(rep, f"parity_{name}.json" if replace_near_ties else f"parity_{name}_free_running.json")
    for k, bound in tol.items():
        assert mx[k] <= bound, (name, k, mx[k], bound, mx)
    return rep


def test_two_stage_flow_conditioned_golden_64():
    """Bounds are fixed numbers over all four rounds (no per-round growth factor, no oracle-sensitivity excuse).
    Measured on MI355X (gpurun_out/parity_traj_fedmlp64.json) over both stem K layouts and three stream-K / tile-order
    variants of the kernels: loss 0.9-1.4e-3, weight norms 1.9-2.4e-3, BN-bias norms 1.8-2.5e-4 (the 32x32 / beta = 0
    golden needed 5e-2 per round there), prototypes 1.5-1.7e-2 and probe logits 2.2-2.7e-2 of their range after 128
    Adam steps from a random init, t off by at most 9-12 of 1024 samples; 1-3 of the 20 picks are near-ties
    (gap <= 3.8e-3 of the row's range) that fall the other way and are replaced by the reference's."""
    _replay_two_stage("traj_fedmlp64", {"loss": 3e-3, "norm": 5e-3, "bn_bias_norm": 1e-3, "proto": 3e-2,
                                        "logits": 4e-2, "t_count": 16})


def test_two_stage_flow_conditioned_golden_64_free_running(monkeypatch):
    """The same replay with NO pick replaced, on the stem layout whose rounding happens to keep every first-round pick
    on the reference's side (FM_STEM_PACKED=0): the whole flow then stays inside the same bounds by itself (measured
    loss 1.4e-3, t off by 11 of 1024) -- the replacement above is not what holds the other runs in."""
    from fedmlp_amd.engine import release_engines
    monkeypatch.setenv("FM_STEM_PACKED", "0")
    release_engines()                      # the stem layout is fixed when an engine is built
    try:
        rep = _replay_two_stage("traj_fedmlp64", {"loss": 3e-3, "norm": 5e-3, "bn_bias_norm": 1e-3, "proto": 3e-2,
                                                  "logits": 4e-2, "t_count": 16}, replace_near_ties=False)
    finally:
        release_engines()
    assert rep["picks_replaced"] == 0


def test_two_stage_flow_c14_golden():
    """14 labels, 3 clients (configs[2] shape).  Measured (both stem layouts): loss 1.6-2.5e-3, norms 1.3-1.5e-3,
    BN-bias norms 1.3-1.5e-4, prototypes 2.3-3.1e-2, probe logits 3.2-3.7e-2, t off by <= 8 of 448; 4 of 18 picks are
    near-ties replaced by the reference's (gap <= 8.9e-3 of range, second stage-2 round)."""
    rep = _replay_two_stage("traj_fedmlp_c14", {"loss": 4e-3, "norm": 4e-3, "bn_bias_norm": 1e-3, "proto": 6e-2,
                                                "logits": 6e-2, "t_count": 14})
    assert rep["picks_total"] > 0
