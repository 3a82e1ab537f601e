"""Round-4 parity additions (goldens made by tests/golden/make_golden.py importing /root/reference in the build container):
  * traj_fedmlp_tail4 / traj_fedmlp_tail1 -- the two-stage FedMLP flow with N = 100 and N = 97 samples per client at bs 32:
                               every pass ends with a batch of 4 / of ONE image per view, normalised by args.batch_size
                               (utils/local_training.py:47-48, 956-959); ChestXray14's 5 889 samples leave exactly such a
                               batch of one at bs 128
  * traj_fixmatch_tails.json -- train_FixMatch (:771-825) with the same two tails
  * effnet_step_224x64.json  -- EfficientNet-B0 stage-1 step at full spatial size and a training-sized batch (bs 64 x 2
                               views x 224x224) through the reference trainer
and, without a stored golden, the full-size shared-mask check: the ORACLE runs on the GPU box's host for one bs 128 x
2 x 224x224 stage-1 step with the engine's ReLU masks, which turns the explanation behind the 3-6e-3 bounds of the
stored-golden tests (test_golden_r2_gpu.py / test_golden_r3_gpu.py) into a checked statement."""
import copy
import os
import time

import numpy as np
import pytest
import torch

from fedmlp_amd import spec
from oracle import steps_ref as R
from tests.helpers import load_golden, make_args, oracle_net, relu_masks_from_engine
from tests.synth import class_lists
from tests.test_golden_r2_gpu import _init_net, _dump, _replay_two_stage
from tests.test_golden_r3_gpu import _effnet_step, BF16_STEP
from tests.test_local_training_gpu import SynthDataset, _norms

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tail", [4, 1])
def test_two_stage_flow_with_a_tail_batch(tail):
    """N = 100 (tail of 4) / N = 97 (tail of ONE image per view, train-mode BatchNorm over a single image) at bs 32, two
    clients, free-running against the reference trainer's trajectory.  Bounds: the fixed numbers of the conditioned goldens
    (test_two_stage_flow_conditioned_golden_64); with ~50 samples per sign the pick bands are wider than at N = 1024 (one
    sample is 2 % of the row), which is what delta says; once a near-tie pick has fallen the other way the two runs train on
    pseudo-labelled sets that differ in one sample of 100 -- 1 % of a round's loss terms, hence 2e-2 on the loss after the split
    (measured 1.26e-2 under the 192-pixel-tile rounding of the conv GEMMs, no split at all under the two roundings before it)."""
    rep = _replay_two_stage(f"traj_fedmlp_tail{tail}",
                            {"loss": 3e-3, "norm": 5e-3, "bn_bias_norm": 1e-3, "proto": 3e-2, "logits": 4e-2, "t_count": 4},
                            {"loss": 2e-2, "norm": 5e-3, "bn_bias_norm": 1e-3, "proto": 6e-2, "logits": 8e-2, "t_count": 12})
    assert rep["picks_total"] > 0


@pytest.mark.parametrize("tail", [4, 1])
def test_fixmatch_round_with_a_tail_batch(tail):
    from tests.helpers import replay_local_update
    LocalUpdate = replay_local_update()
    g = load_golden("traj_fixmatch_tails.json")[f"tail{tail}"]
    C, N = g["C"], g["N"]
    args = make_args(n_classes=C, n_clients=1, seed=g["init_seed"])
    ds = SynthDataset(N, C, g["hw"], g["data_seed"], True)
    pos, neg = class_lists(ds.targets, C)
    net = _init_net(args, g["bn_seed"])
    sd = net.state_dict()
    sd["fc.weight"] = sd["fc.weight"] * g["fc_scale"]
    net.load_state_dict(sd)
    loc = LocalUpdate(args, 0, ds, list(range(N)), pos, neg, active_class_list=[0])
    np.testing.assert_allclose(loc.loss_w, g["loss_w"], rtol=0)
    np.testing.assert_allclose(loc.loss_w_unknown, g["loss_w_unknown"], rtol=0)
    loc.order_queue.append(g["order"])
    out = loc.train_FixMatch(0, net)
    got = _norms(out[0])
    rep = {"loss": float(out[1]), "loss_ref": g["loss"], "loss_rel_err": abs(out[1] - g["loss"]) / abs(g["loss"])}
    worst = {"bn_bias": 0.0, "other": 0.0}
    for k, w in g["norms"].items():
        if "num_batches" in k:
            assert abs(got[k] - w) < 0.5, k
            continue
        kind = "bn_bias" if (k.endswith(".bias") and not k.startswith("fc.")) else "other"
        worst[kind] = max(worst[kind], abs(got[k] - w) / (abs(w) + 1e-12))
    rep["norm_rel_err"] = worst
    _dump(rep, f"parity_traj_fixmatch_tail{tail}.json")
    # 4 Adam steps from a conditioned init (non-trivial BN affine): the bounds of the conditioned goldens
    assert rep["loss_rel_err"] < 2e-3, rep
    assert worst["other"] < 1e-3 and worst["bn_bias"] < 1e-3, rep


def test_effnet_stage1_step_full_size_training_batch_fp32():
    """bs 64 x 2 views x 224x224 through the reference trainer (same bounds as the bs-16 / bs-256 goldens)"""
    _effnet_step("effnet_step_224x64.json", "fp32", {"loss": 2e-6, "worst": 5e-4, "median": 2e-5})


def test_effnet_stage1_step_full_size_training_batch_bf16():
    _effnet_step("effnet_step_224x64.json", "bf16", BF16_STEP)


STEM_BOUND = 2e-4        # measured with the shared stem decisions: conv1.weight 5.0e-6, bn1.weight 4.6e-5, bn1.bias 8.3e-5 (12 of 2.1e8 pool choices, 293 of 5.9e8 ReLU masks differ)


def test_stage1_step_at_the_benchmarked_size_against_the_oracle_with_shared_relu_masks():
    """bs 128 x 2 views x 224x224, C = 5 (what bench.py times).  The stored-golden tests hold 3-6e-3 per gradient tensor
    there and explain it by ReLU masks: of 5.9e8 ReLU inputs a few hundred lie within fp32 rounding distance of zero and
    fall differently under the two summation orders.  Here the oracle runs on this host with the ENGINE's masks in its
    backward (tests/helpers.relu_masks_from_engine; its forward and the loss are untouched), the differing positions are
    counted, and what is left -- the backward arithmetic itself at full size -- must agree to 2e-4 of each tensor's max,
    the bound the 64x64 shared-mask tests hold (tests/test_engine_gpu.py).  Round 5: the stem is inside the statement too -- the
    engine hands over its stem ReLU mask and its max-pool choices (fm_debug_stem_masks; the oracle's max-pool backward routes each
    gradient to the position the engine chose), so conv1.weight, bn1.weight and bn1.bias hold the same 2e-4 as the other 59
    tensors (measured 5.0e-6 / 4.6e-5 / 8.3e-5; without the hand-over they sat at 1.7e-3 / 4.4e-5 / 9.0e-4, which was the
    choices, not the arithmetic).  For conv1.weight (3.2e6-term sums) the yardstick stays the same oracle step in FLOAT64: the
    engine must be as close to that as the fp32 oracle is."""
    from fedmlp_amd.engine import Engine
    C, B, hw = 5, 128, 224
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    g = torch.Generator().manual_seed(4242)
    x1 = torch.randn((B, 3, hw, hw), generator=g)
    x2 = x1 + 0.1 * torch.randn((B, 3, hw, hw), generator=g)
    y = (torch.rand((B, C), generator=g) < 0.3).float()
    act, neg = [0], [1, 2, 3, 4]
    y[:, 1:] = 0.0
    net0 = oracle_net(C, 1037)
    flat, cnt = spec.state_dict_to_flat("Resnet18", C, net0.state_dict())
    eng = Engine("Resnet18", C, hw, hw, 2 * B)
    try:
        eng.set_state(flat, cnt)
        eng.teacher_snapshot()
        eng.adam_reset(3e-5)
        lo = torch.zeros(1, device=eng.device)
        eng.step_stage1(x1.to(eng.device), x2.to(eng.device), y.to(eng.device), [1.0, 0, 0, 0, 0], 1, B, lo)
        got_loss = lo.item()
        gsd = spec.flat_to_state_dict("Resnet18", C, eng.debug_get_grads(), np.zeros(eng.ni, np.int64))
        rm = relu_masks_from_engine(eng, 2, B, stem=True)
    finally:
        eng.close()

    def oracle_step(dtype):
        net = copy.deepcopy(net0).to(dtype)
        glob = copy.deepcopy(net).eval()
        net.train()
        a, b, yy = x1.to(dtype), x2.to(dtype), y.to(dtype)
        rm.calls = rm.flips = rm.pool_calls = rm.pool_flips = 0
        with rm:
            _, z1 = net(a); _, z2 = net(b)
            with torch.no_grad():
                _, g1 = glob(a); _, g2 = glob(b)
            loss, _, _ = R.loss_stage1(z1, z2, g1, g2, yy, act, neg, B, 1)
            loss.backward()
        rep_pool["flips"] = int(rm.pool_flips)
        return loss.item(), {k: p.grad.double().numpy() for k, p in net.named_parameters()}, int(rm.flips), rm.calls

    rep_pool = {}
    t0 = time.perf_counter()
    loss32, g32, flips, calls = oracle_step(torch.float32)
    pool_flips = rep_pool["flips"]
    oracle_s = time.perf_counter() - t0
    errs = {k: float(np.abs(gsd[k] - w).max() / (np.abs(w).max() + 1e-12)) for k, w in g32.items()}
    worst = max(errs.items(), key=lambda kv: kv[1])
    # conv1.weight, bn1.weight, bn1.bias sit at or below the stem's ReLU / max-pool, whose mask and argmax are each side's own
    STEM = ("conv1.weight", "bn1.weight", "bn1.bias")
    rest = {k: v for k, v in errs.items() if k not in STEM}
    worst_rest = max(rest.items(), key=lambda kv: kv[1])
    n_relu = 2 * B * (64 * 112 * 112 + 4 * 64 * 56 * 56 + 4 * 128 * 28 * 28 + 4 * 256 * 14 * 14 + 4 * 512 * 7 * 7)
    rep = {"loss": got_loss, "loss_oracle": loss32, "loss_rel_err": abs(got_loss - loss32) / abs(loss32),
           "relu_inputs": n_relu, "mask_flips": flips, "relu_calls": calls, "maxpool_choices_differing": pool_flips,
           "stem_tensors": {k: errs[k] for k in STEM},
           "worst_grad_tensor": worst[0], "worst_grad_rel_to_max": worst[1],
           "worst_below_the_stem": list(worst_rest), "median_grad_rel_to_max": float(np.median(list(errs.values()))),
           "top5": sorted(errs.items(), key=lambda kv: -kv[1])[:5], "oracle_seconds": round(oracle_s, 1),
           "threads": torch.get_num_threads()}
    # float64 oracle (same masks handed over): where do the two fp32 implementations stand against it?
    try:
        import psutil
        room = psutil.virtual_memory().available > (110 << 30)
    except Exception:
        room = False
    if room:
        t0 = time.perf_counter()
        loss64, g64, _, _ = oracle_step(torch.float64)
        e_eng = {k: float(np.abs(gsd[k] - w).max() / (np.abs(w).max() + 1e-300)) for k, w in g64.items()}
        e_o32 = {k: float(np.abs(g32[k] - w).max() / (np.abs(w).max() + 1e-300)) for k, w in g64.items()}
        rep["float64"] = {"loss": loss64, "seconds": round(time.perf_counter() - t0, 1),
                          "conv1.weight": {"engine_vs_f64": e_eng["conv1.weight"], "oracle_f32_vs_f64": e_o32["conv1.weight"]},
                          "bn1.bias": {"engine_vs_f64": e_eng["bn1.bias"], "oracle_f32_vs_f64": e_o32["bn1.bias"]},
                          "worst_engine_vs_f64": list(max(e_eng.items(), key=lambda kv: kv[1])),
                          "worst_oracle_f32_vs_f64": list(max(e_o32.items(), key=lambda kv: kv[1])),
                          "median_engine_vs_f64": float(np.median(list(e_eng.values()))),
                          "median_oracle_f32_vs_f64": float(np.median(list(e_o32.values()))),
                          # round 6: the whole trace, tensor by tensor in backward order -- where along the
                          # network does the engine's distance from float64 leave the fp32 oracle's?
                          "per_tensor": {k: [e_eng[k], e_o32[k]] for k in reversed(list(g64.keys()))}}
    # the same record per product form (FM_MFMA_SPLIT=0: fp32 matrix pipe, 9: nine partial products): which part of the distance
    # belongs to the six-product form and which to the matrix pipe's own accumulation
    form = os.environ.get("FM_MFMA_SPLIT")
    if os.environ.get("FM_PLANES") == "0":
        form = (form or "6") + "_fp32_operand_kernels"
    _dump(rep, "parity_step_full_shared_masks.json" if not form else f"parity_step_full_shared_masks_products{form}.json")
    assert rep["loss_rel_err"] < 1e-5, rep
    assert calls == 4 * 17                                  # 2 train-mode + 2 teacher forwards x 17 ReLUs
    assert flips <= 2e-6 * n_relu, rep                      # measured: 266 of 5.9e8
    assert worst_rest[1] < 2e-4, rep                        # measured: 3.5e-5 (layer1.0.bn2.bias), median 6e-6
    # the stem's three tensors: 1.1e-3 / 4e-5 / 1.7e-5 while the stem conv ran on the fp32 pipe (whose K = 147 sums come out
    # nearly bit-equal to torch's, so the two stem masks hardly differ), 1.7e-3 / 4.4e-5 / 9.0e-4 in the six-product form
    # (closer to float64, tests/test_kernels_gpu.py, but 1e-7 away from torch's fp32 values like every other layer: two or
    # three of bn1's 2.1e8 ReLU inputs change sign, and ONE of them is 5.6e-4 of a bias gradient that is a 3.2e6-term sum
    # of cancelling signs)
    # round 5: the stem's ReLU mask and max-pool choices are handed over too (fm_debug_stem_masks), so its three tensors hold the
    # bound of the other 59
    for k in STEM:
        assert errs[k] < STEM_BOUND, (k, rep)
    if room:
        f = rep["float64"]["conv1.weight"]
        assert f["engine_vs_f64"] < 3.0 * f["oracle_f32_vs_f64"] + 2e-4, rep


def _grad_record(grads):
    """{key: {norm, sum, head, absmax}} of a dict of numpy gradients (the record format of tests/golden/make_golden.py)"""
    out = {}
    for k, g in grads.items():
        g = np.asarray(g, np.float64)
        out[k] = {"norm": float(np.linalg.norm(g)), "sum": float(g.sum()), "head": [float(v) for v in g.ravel()[:3]],
                  "absmax": float(np.abs(g).max())}
    return out


# ---- bf16 (BASELINE configs[4]): whose error is it? ------------------------------------------------------------------------
# The reference never runs reduced precision, so the bf16 tolerance is this build's and has to be defended.  Yardstick: the fp32
# oracle with a bf16 rounding inserted at exactly the engine's storage points (oracle/efficientnet_ref.py, storage="bf16":
# activations and their gradients where the engine stores them, bf16 MFMA weight operands) -- an INDEPENDENT implementation of the
# same storage rule, run on this host.  Three sides on the goldens' own inputs and initial state: the reference trainer's fp32 step
# (R, the committed golden), the storage-emulating oracle (O) and the engine in bf16 mode (E).  Statement: E is no further from R
# than O is -- per summary statistic, and on the engine's worst tensor O is off by a comparable amount (that tensor's gradient is
# ill-conditioned under bf16 storage, whoever implements it).
@pytest.mark.parametrize("gname", ["effnet_step_bs256.json", "effnet_step.json", "effnet_step_224x64.json"])
def test_effnet_bf16_deviation_is_the_storage_rule_not_the_engine(gname):
    from fedmlp_amd.engine import Engine
    from fedmlp_amd.spec import state_dict_to_flat, flat_to_state_dict
    from oracle.efficientnet_ref import EfficientNetB0Ref
    from tests.helpers import grad_errors
    from tests.synth import synth_arrays
    g = load_golden(gname)
    C, N, hw, M = g["C"], g["N"], g["hw"], "Efficient_b0"
    args = make_args(n_classes=C, n_clients=1, batch_size=g["bs"], seed=g["init_seed"], pretrained=0, model=M, feature_dim=1280)
    hnet = _init_net(args, g["bn_seed"])
    targets, x1, x2 = synth_arrays(N, C, hw, g["data_seed"], True)
    y = targets.copy(); y[:, 1:] = 0.0
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    net = EfficientNetB0Ref(C, storage="bf16")
    net.load_state_dict({k: torch.as_tensor(np.asarray(v)) for k, v in hnet.state_dict().items()})
    glob = copy.deepcopy(net).eval()
    net.train()
    X1, X2, Y = torch.from_numpy(x1), torch.from_numpy(x2), torch.from_numpy(y)
    t0 = time.perf_counter()
    _, z1 = net(X1); _, z2 = net(X2)
    with torch.no_grad():
        _, g1 = glob(X1); _, g2 = glob(X2)
    loss, _, _ = R.loss_stage1(z1, z2, g1, g2, Y, [0], list(range(1, C)), g["bs"], 1)
    loss.backward()
    oracle_s = time.perf_counter() - t0
    o16 = {k: p.grad.numpy() for k, p in net.named_parameters()}
    ref = g["stage1"]["grads"]

    def summary(got):
        k_bad, worst, med = grad_errors(got, ref)
        return {"worst_tensor": k_bad, "worst": worst, "median": med, "p90": grad_errors.p90}, dict(grad_errors.all)

    s_o, e_o = summary(o16)
    eng = Engine(M, C, hw, hw, 2 * N, precision="bf16")
    try:
        eng.stochastic = False
        flat, cnt = state_dict_to_flat(M, C, hnet.state_dict())
        eng.set_state(flat, cnt)
        eng.adam_reset(3e-5)
        eng.teacher_snapshot()
        lo = torch.zeros(1, device=eng.device)
        eng.step_stage1(X1.to(eng.device), X2.to(eng.device), Y.to(eng.device), [1.0] + [0.0] * (C - 1), 1, g["bs"], lo)
        got = flat_to_state_dict(M, C, eng.debug_get_grads(), np.zeros(eng.ni, np.int64))
    finally:
        eng.close()
    s_e, e_e = summary(got)
    top = sorted(e_e.items(), key=lambda kv: -kv[1])[:8]
    rep = {"loss": {"engine_bf16": lo.item(), "oracle_bf16_storage": loss.item(), "reference_fp32": g["stage1"]["loss"]},
           "engine_vs_reference": s_e, "emulating_oracle_vs_reference": s_o,
           "engine_worst_tensors": [[k, round(v, 5), round(e_o[k], 5)] for k, v in top],     # [tensor, engine err, oracle err]
           "oracle_seconds": round(oracle_s, 1)}
    lr, lref = abs(lo.item() - g["stage1"]["loss"]), abs(loss.item() - g["stage1"]["loss"])
    rep["loss"]["engine_rel_dev"], rep["loss"]["oracle_rel_dev"] = lr / abs(g["stage1"]["loss"]), lref / abs(g["stage1"]["loss"])
    _dump(rep, f"parity_effnet_bf16_three_sides_{gname[:-5]}.json")
    assert lr < 2e-3 * abs(g["stage1"]["loss"]), rep
    for k in ("median", "p90"):
        assert s_e[k] <= 1.5 * s_o[k] + 5e-3, (k, rep)
    # the worst tensor: the independent implementation is off by a comparable amount on it, or the engine is within the general bound
    kw = s_e["worst_tensor"]
    assert s_e["worst"] < 0.3 or e_o[kw] > 0.25 * s_e["worst"], rep
    assert sum(1 for v in e_e.values() if v > 0.3) <= max(1, sum(1 for v in e_o.values() if v > 0.3)), rep
