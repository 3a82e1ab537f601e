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


def _boundary(sim, k, top):
    """midpoint between the k-th pick and the first non-pick of a similarity row (descending for the top picks,
    ascending for the bottom picks); +-inf when the selection takes nothing / everything"""
    o = np.sort(sim)[::-1] if top else np.sort(sim)
    if k <= 0:
        return np.inf if top else -np.inf
    if k >= len(o):
        return -np.inf if top else np.inf
    return 0.5 * (float(o[k - 1]) + float(o[k]))


def _check_picks(rep, mx, P, rnd, i, k, cls, log, got_clean, got_noise, args, delta):
    """Discrete picks of one tagging call against the REFERENCE's similarity row (recorded by make_golden.py from the
    reference's own run).  Nothing is replaced: the engine's picks stand.  With delta = the similarity noise allowed
    between two implementations (as a fraction of the reference row's range):
      * every reference pick whose reference margin to the selection boundary exceeds delta must be picked here too,
      * nothing may be picked here that the reference ranks more than delta behind its boundary,
      * the pick COUNTS int(thr * #(sim >= 0)) / int(thr * #(sim < 0)) may differ only as far as the samples within
        delta of zero can move the sign counts,
      * on the samples both pools hold, the similarities themselves agree to delta (reported as mx["sim"]).
    Picks inside the band that fall the other way are counted in rep["picks_differing"]."""
    s_ref = P[f"r{rnd}_c{i}_k{k}_sim"].astype(np.float64)
    pool_ref = [int(v) for v in P[f"r{rnd}_c{i}_k{k}_pool"]]
    if np.isnan(s_ref).all():                       # no prototype for this class: the reference selects nothing either
        assert not got_clean and not got_noise
        return
    rng = float(np.nanmax(s_ref) - np.nanmin(s_ref))
    ref_of = dict(zip(pool_ref, s_ref))
    eng_of = dict(zip(log["pool_idx"], log["sim"].astype(np.float64)))
    common = [v for v in pool_ref if v in eng_of]
    if common:
        mx["sim"] = max(mx.get("sim", 0.0), max(abs(eng_of[v] - ref_of[v]) for v in common) / rng)
    nz = int((np.abs(s_ref) <= delta * rng).sum())
    npos, nneg = int((s_ref >= 0).sum()), int((s_ref < 0).sum())
    kt, kb = int(args.clean_threshold * npos), int(args.noise_threshold * nneg)
    if kt == 0:
        kb = 0          # the reference keeps no noise pick for a class without a clean pick (:1076 tests the clean list twice)
    if len(pool_ref) == len(log["pool_idx"]):      # same pool size: the counts are bounded by the sign-flip band
        ok_t = {int(args.clean_threshold * n) for n in range(max(0, npos - nz), npos + nz + 1)}
        ok_b = {int(args.noise_threshold * n) for n in range(max(0, nneg - nz), nneg + nz + 1)}
        if 0 in ok_t:
            ok_b.add(0)
        assert len(got_clean) in ok_t and len(got_noise) in ok_b, (rnd, i, cls, len(got_clean), kt, len(got_noise), kb)
    for top, got, kk in ((True, got_clean, kt), (False, got_noise, kb)):
        b = _boundary(s_ref, kk, top)
        sgn = 1.0 if top else -1.0
        want = [v for v in pool_ref if sgn * (ref_of[v] - b) > 0]          # the reference's picks of this call
        for v in want:
            margin = sgn * (ref_of[v] - b) / rng
            if v in eng_of and margin > delta:
                assert v in got, ("reference pick with a clear margin is missing", rnd, i, cls, v, margin)
        for v in got:
            if v in ref_of:
                behind = -sgn * (ref_of[v] - b) / rng
                mx["pick_behind"] = max(mx["pick_behind"], behind)
                assert behind <= delta, ("pick far behind the reference's boundary", rnd, i, cls, v, behind)
        rep["picks_total"] += len(want)
        rep["picks_differing"] += len(set(want) - set(got))


def reference_band():
    """The similarity band of the ResNet-18 two-stage replays, DERIVED from how far the reference moves from itself: the committed
    record tests/golden/oracle_bands.json (tests/golden/oracle_bands.py re-runs the reference's own traj_fedmlp64 flow under
    other fp32 summation orders: 3 threads instead of 8, torch's native convolutions instead of oneDNN) holds, per variant, the
    largest similarity difference to the committed golden as a fraction of the row's range -- 1.5e-2 / 2.4e-2 (first / later
    stage-2 rounds) for the thread count alone, 2.1e-2 / 3.4e-2 (and 6 of 30 picks changed) for the other convolution kernels.
    Two implementations that each sit one such band from the exact trajectory differ by up to two: delta = 2 x band + 1e-2."""
    with open(os.path.join(GOLDEN, "oracle_bands.json")) as f:
        b = json.load(f)["traj_fedmlp64"]
    first = max(v["sim_first_round"] for v in b.values())
    later = max(v["sim_later_rounds"] for v in b.values())
    return (2 * first + 1e-2, 2 * later + 1e-2)


def _replay_two_stage(name, tol, tol_after_split, delta=None, model="Resnet18", precision="fp32"):
    """The FedMLP two-stage flow (stage 1, prototype pass, tagging + selection, stage 2, FedAvg*) FREE-RUNNING against a
    golden trajectory of the reference: the engine's own picks train the following rounds, nothing is replaced.
    tol: fixed bounds {loss, norm, bn_bias_norm, proto, logits, t_count} that hold while every pick so far equals the
    reference's (the two runs are the same experiment); tol_after_split: the bounds from the first round in which a
    near-tie pick fell the other way (two experiments that differ in a few pseudo-labelled samples out of ~1000).
    delta = similarity band of _check_picks in the first / in later stage-2 rounds; the similarities themselves must agree to
    delta[0] while the runs are the same experiment (and to 2 * delta[1] afterwards).  ResNet-18 from a random init: after
    the two stage-1 rounds (64 Adam steps) the engine's and the reference's similarity rows differ by 1.2-3.3e-2 of the row's
    range, depending on nothing but fp32 rounding: measured on the C = 14 golden under three roundings of the engine's conv
    GEMMs -- fp32 matrix pipe 1.5e-2, bf16 partial products with 256-pixel tiles (and BN partial sums) on the 64-channel layers
    1.2e-2, the same with 192-pixel tiles 3.3e-2 (the tail-of-4 golden: 0.6 / 1.2e-2 / a pick 4.0e-2 behind the boundary) --
    the chaos amplification of fp32 rounding that profiles/HISTORY.md quantifies.  Round 5: the band is no longer a chosen number but
    reference_band(): twice what the REFERENCE moves under another summation order, + 1e-2 (5.1e-2, 7.8e-2 from the committed
    record: the engine's 1.2-3.3e-2 sits inside the reference's own 2.1-3.4e-2)."""
    if delta is None:
        delta = reference_band()
    from tests.helpers import replay_local_update
    LocalUpdate = replay_local_update()      # LocalUpdate + recorded batch orders / tagging log (tests/helpers.py)
    from fedmlp_amd.fedavg import FedAvg, FedAvg_tao, FedAvg_proto
    g = load_golden(name + ".json")
    P = np.load(os.path.join(GOLDEN, name + "_protos.npz"))
    C, n_cl, N, S1 = g["C"], g["n_clients"], g["N"], g["S1"]
    args = make_args(n_classes=C, n_clients=n_cl, rounds_FedMLP_stage1=S1, seed=g["init_seed"], pretrained=0, model=model,
                     feature_dim=spec.FEATURE_DIM[model], precision=precision,
                     clean_threshold=g.get("clean_threshold", 0.005), noise_threshold=g.get("noise_threshold", 0.01))
    ds = SynthDataset(n_cl * N, C, g["hw"], g["data_seed"], True, g.get("p_pos", 0.3))
    pos, neg = class_lists(ds.targets, C)
    netglob = _init_net(args, g.get("bn_seed"))
    bnstats = os.path.join(GOLDEN, name + "_bnstats.npz")
    if os.path.exists(bnstats):                 # calibrated running statistics of the golden's initial net (make_golden.py)
        sd = netglob.state_dict()
        for k, v in np.load(bnstats).items():
            sd[k] = torch.from_numpy(v)
        netglob.load_state_dict(sd)
    locs = [LocalUpdate(args, i, ds, g["users"][i], pos, neg, active_class_list=[i]) for i in range(n_cl)]
    if model == "Efficient_b0":                 # the reference trainer passes no drop-connect / dropout multipliers
        locs[0]._bind(netglob, "image_aug_1").stochastic = False
    tao, Prototype = [0] * C, []
    neg_lists, act_lists = g["neg_lists"], g["act_lists"]
    keys = ("loss", "norm", "bn_bias_norm", "se_bias_norm", "proto", "logits", "t_count")
    rep = {"max": {k: 0.0 for k in keys}, "max_after_split": {k: 0.0 for k in keys}, "picks_differing": 0,
           "picks_total": 0, "split_round": None}
    rep["max"].update(sim=0.0, pick_behind=-1.0)
    rep["max_after_split"].update(sim=0.0, pick_behind=-1.0)
    mx = rep["max"]
    cur = mx                                       # the record the continuous deviations go to

    def cmp_norms(got, want):
        for k, w in want.items():
            if "num_batches" in k:
                assert abs(got[k] - w) < 0.5, k
                continue
            # EfficientNet's squeeze-excite conv biases start at ZERO (their norm after a few dozen Adam steps is a sum of +-lr
            # steps whose signs follow gradient noise): their own kind, apart from the BatchNorm biases of a conditioned init
            kind = ("se_bias_norm" if "_se_" in k else "bn_bias_norm") if (k.endswith(".bias") and not k.startswith("fc.")) else "norm"
            cur[kind] = max(cur[kind], abs(got[k] - w) / (abs(w) + 1e-12))

    for rnd, r in enumerate(g["rounds"]):
        w, taos, protos = [], [], []
        for i in range(n_cl):
            if rnd < S1:
                locs[i].order_queue.append(r["train_orders"][i])
                a1 = (None, None) if rnd < S1 - 1 else (neg_lists[i], act_lists[i])
            else:
                locs[i].order_queue += [r["feat_orders"][i], r["train_orders"][i]]
                a1 = (neg_lists[i], act_lists[i])
                before = [list(l) for l in locs[i].traindata_idx] if rnd > S1 else None
                nlog = len(locs[i].tagging_log)
            ret = locs[i].train_FedMLP(rnd, tao, Prototype, None, a1[0], a1[1], net=copy.deepcopy(netglob))
            if rnd >= S1:
                # the tagging of this call happened before its stage-2 training: judge the picks, then the continuous
                # quantities of the round go to the record that matches the state of the experiment
                logs = locs[i].tagging_log[nlog:]
                by_cls = {l["cls"]: l for l in logs}
                for k, cls in enumerate(neg_lists[i]):
                    new_c = locs[i].traindata_idx[2 * k][len(before[2 * k]) if before else 0:]
                    new_n = locs[i].traindata_idx[2 * k + 1][len(before[2 * k + 1]) if before else 0:]
                    if cls in by_cls:
                        _check_picks(rep, cur, P, rnd, i, k, cls, by_cls[cls], new_c, new_n, args,
                                     delta[0] if rnd == S1 else delta[1])
                    else:
                        assert not new_c and not new_n
                same = [sorted(a) for a in locs[i].traindata_idx] == [sorted(b) for b in r["traindata_idx"][i]]
                if not same and rep["split_round"] is None:
                    rep["split_round"] = rnd
                    cur = rep["max_after_split"]
            cur["loss"] = max(cur["loss"], abs(ret[1] - r["loss"][i]) / abs(r["loss"][i]))
            cmp_norms(_norms(ret[0]), r["norms"][i])
            if rnd == 0:
                assert ret[4] == neg_lists[i] and ret[5] == act_lists[i]
            w.append(copy.deepcopy(ret[0]))
            if len(ret) == 8:
                taos.append(ret[6]); protos.append(ret[7])
                cur["t_count"] = max(cur["t_count"], float(np.abs(ret[6] - P[f"r{rnd}_c{i}_t"]).max() * N))
                want = P[f"r{rnd}_c{i}_proto"]
                assert np.array_equal(np.isnan(ret[7].numpy()), np.isnan(want))
                cur["proto"] = max(cur["proto"], float(np.nanmax(np.abs(ret[7].numpy() - want)) / np.nanmax(np.abs(want))))
        netglob.load_state_dict(FedAvg(w, [N] * n_cl))
        if rnd >= S1 - 1:
            tao = FedAvg_tao(taos, [N] * n_cl, g["class_negative_client_list"])
            Prototype = FedAvg_proto(protos, [N] * n_cl, g["class_active_client_list"])
            want = P[f"r{rnd}_glob_proto"]
            assert np.array_equal(np.isnan(Prototype.numpy()), np.isnan(want))              # NaN rows
            cur["proto"] = max(cur["proto"], float(np.nanmax(np.abs(Prototype.numpy() - want)) / np.nanmax(np.abs(want))))
            cur["t_count"] = max(cur["t_count"], float(np.abs(tao - np.array(r["tao"])).max() * N))
        cmp_norms(_norms(netglob.state_dict()), r["glob_norms"])
        netglob.eval()
        _, z = netglob(ds.x1[:4])
        want = np.array(r["probe_logits"])
        cur["logits"] = max(cur["logits"], float(np.abs(z.cpu().numpy() - want).max() / np.abs(want).max()))
        rep.setdefault("running_max_after_round", []).append({"same_experiment": dict(mx), "after_split": dict(rep["max_after_split"])})
    _dump(rep, f"parity_{name}.json" if precision == "fp32" else f"parity_{name}_{precision}.json")
    for k, bound in tol.items():
        assert mx[k] <= bound, (name, k, mx[k], bound, rep)
    for k, bound in tol_after_split.items():
        assert rep["max_after_split"][k] <= bound, (name, "after the split", k, rep["max_after_split"][k], bound, rep)
    assert mx["sim"] <= delta[0] and rep["max_after_split"]["sim"] <= 2 * delta[1], (name, mx["sim"], rep["max_after_split"]["sim"])
    return rep


def test_two_stage_flow_conditioned_golden_64():
    """Free-running on the library defaults; bounds are fixed numbers over the rounds (no per-round growth factor).
    While the picks equal the reference's: loss 3e-3, weight norms 5e-3, BN-bias norms 1e-3, prototypes 3e-2, probe
    logits 4e-2 of their range after 128 Adam steps from a random init, t within 16 of 1024 samples.  One boundary of
    this golden is a near-tie IN THE REFERENCE (client 1, class 0, first stage-2 round: the 4th and 5th most similar
    samples are 1.2e-3 of the row's range apart, `make_golden.py` prints every margin): an implementation with other
    rounding may take the other one, after which the two runs train on pseudo-labelled sets that differ in one sample
    of 1024 and are compared with the `after the split` bounds (t moves by tens of samples: its thresholds L / U cut
    through the bulk of a barely trained classifier's probabilities)."""
    rep = _replay_two_stage("traj_fedmlp64",
                            {"loss": 3e-3, "norm": 5e-3, "bn_bias_norm": 1e-3, "proto": 3e-2, "logits": 4e-2, "t_count": 16},
                            {"loss": 1e-2, "norm": 5e-3, "bn_bias_norm": 1e-3, "proto": 6e-2, "logits": 8e-2, "t_count": 96})
    assert rep["picks_total"] == 20


def test_two_stage_flow_c14_golden():
    """14 labels, 3 clients (configs[2] shape), free-running: NaN prototype rows for the 11 classes nobody annotates,
    tagging only where a prototype exists."""
    rep = _replay_two_stage("traj_fedmlp_c14",
                            {"loss": 4e-3, "norm": 4e-3, "bn_bias_norm": 1e-3, "proto": 6e-2, "logits": 6e-2, "t_count": 14},
                            {"loss": 1.2e-2, "norm": 4e-3, "bn_bias_norm": 1e-3, "proto": 1e-1, "logits": 1e-1, "t_count": 64})
    assert rep["picks_total"] > 0
