"""Pins the CPU oracle (oracle/) against goldens captured from the imported
reference (tests/golden/make_golden.py).  CPU only."""
import copy

import numpy as np
import pytest
import torch

from oracle import steps_ref as R
from tests.helpers import (load_golden, make_args, oracle_net, data_dict, norms_of,
                           assert_norms_close, GOLDEN)
from tests.synth import class_lists

torch.set_num_threads(8)


@pytest.fixture(scope="module")
def kat():
    return load_golden("kat.json")


def test_fedavg_kat(kat):
    g = kat["fedavg"]
    w = [{k: torch.tensor(v, dtype=torch.int64 if "num_batches" in k else torch.float32)
          for k, v in wi.items()} for wi in g["w"]]
    out = R.fedavg(w, g["lens"])
    for k, v in g["out"].items():
        assert str(out[k].dtype) == g["out_dtype"][k]
        np.testing.assert_array_equal(out[k].numpy(), np.array(v, dtype=out[k].numpy().dtype))


def test_fedavg_tao_proto_kat(kat):
    g = kat["fedavg_tao"]
    out = R.fedavg_tao([np.array(v) for v in g["t"]], g["weight"], g["clients"])
    np.testing.assert_array_equal(out, np.array(g["out"]))
    g = kat["fedavg_proto"]
    out = R.fedavg_proto([torch.tensor(p) for p in g["P"]], g["weight"], g["clients"]).numpy()
    want = np.array([[np.nan if x is None else x for x in r] for r in g["out"]], dtype=np.float32)
    np.testing.assert_array_equal(out, want)      # NaN rows for the class with no client
    assert np.isnan(out[4]).all() and np.isnan(out[5]).all()


def test_cosine_kat(kat):
    g = kat["cosine"]
    sim = R.cosine_diff(torch.tensor(g["f"]), torch.tensor(g["p0"]), torch.tensor(g["p1"]))
    np.testing.assert_array_equal(sim.numpy(), np.array(g["sim"], dtype=np.float32))
    f3 = torch.tensor([[1., 2, 3], [0, -1, .5], [2, 0, 0]])
    sim = R.cosine_diff(f3, torch.tensor([1., 0, 0]), torch.tensor([0., 1, 1]))
    np.testing.assert_array_equal(sim.numpy(),
                                  np.array(kat["cosine_survey"]["sim"], dtype=np.float32))
    np.testing.assert_allclose(sim.numpy(), [-0.6776499748, 0.3162277639, 1.0], rtol=1e-6)


def test_topk_kat(kat):
    g = kat["topk"]
    for n in range(9):
        assert R.max_m_indices(g["lst"], n) == g["max"][str(n)]
        assert R.min_n_indices(g["lst"], n) == g["min"][str(n)]
    assert R.max_m_indices([.1, .5, .5, -1], 2) == [1, 2]
    assert R.min_n_indices([.1, -1, -1, 3], 2) == [1, 2]


def test_bce_probs_kat(kat):
    g = kat["bce_probs"]
    out = R.bce_on_probs(torch.tensor(g["p"]), torch.tensor(g["y"]))
    np.testing.assert_array_equal(out.numpy(), np.array(g["out"], dtype=np.float32))
    assert out.max().item() == 100.0          # log clamp


def test_dataset_split_kat(kat):
    g = kat["dataset_split"]
    t = np.array(g["targets"], dtype=np.float32)
    ym = R.mask_targets(t, g["idxs"], g["active"], g["class_neg_idx"])
    np.testing.assert_array_equal(ym, np.array(g["masked"], dtype=np.float32))
    assert R.class_counts(t, g["idxs"]) == g["counts"]
    y, d = R.pseudo_targets(t, g["idxs"], g["active"], g["negative"], g["traindata_idx"])
    np.testing.assert_array_equal(y, np.array(g["pseudo_y"], dtype=np.float32))
    np.testing.assert_array_equal(d, np.array(g["pseudo_distill"], dtype=np.float32))


def test_loss_heads_survey_kat(kat):
    g = kat["loss_survey"]
    z1, z2, g1, g2 = (torch.tensor(g[k], dtype=torch.float32) for k in ("z1", "z2", "g1", "g2"))
    y = torch.tensor(g["y"], dtype=torch.float32)
    tot, sup, dis = R.loss_stage1(z1, z2, g1, g2, y, g["active"], g["negative"], g["bs_norm"], 1)
    assert abs(sup.item() - g["stage1_sup"]) < 2e-7
    assert abs(dis.item() - g["stage1_dis"]) < 2e-7
    assert abs(tot.item() - g["stage1_total"]) < 2e-7
    l = R.loss_train(z1, y, g["pos_weight"], g["bs_norm"], 4)
    assert abs(l.item() - g["train"]) < 2e-7
    y2 = y.clone(); y2[0, 2] = 1
    l = R.loss_stage2(z1, y2, torch.tensor(g["distill_cls"], dtype=torch.float32))
    assert abs(l.item() - g["stage2"]) < 2e-7


def test_find_indices_semantics(kat):
    g = kat["find_indices"]
    where = {v: j for j, v in enumerate(g["a"])}
    assert [where[float(v)] for v in g["b"]] == g["out"]


# ------------------------------------------------------------------ trajectories
def test_traj_train_config1():
    """BASELINE config 1: 2 clients, ResNet-18, warm-up BCE only, bs 32 (reduced to 32x32)."""
    g = load_golden("traj_train.json")
    C, n_cl, N = g["C"], g["n_clients"], g["N"]
    args = make_args(n_classes=C, n_clients=n_cl)
    data = data_dict(n_cl * N, C, g["hw"], g["data_seed"], False)
    _, neg = class_lists(data["targets"], C)
    glob = _perturb_bn(oracle_net(C, g["init_seed"]), g.get("bn_seed"))
    clients = [R.RefClient(args, i, data, g["users"][i], neg, [i]) for i in range(n_cl)]
    for i in range(n_cl):
        np.testing.assert_allclose(clients[i].loss_w, g["loss_w"][i], rtol=0)
    for rnd, r in enumerate(g["rounds"]):
        w = []
        for i in range(n_cl):
            sd, loss, _ = clients[i].train(copy.deepcopy(glob), r["orders"][i])
            assert abs(loss - r["loss"][i]) <= 1e-5 * abs(r["loss"][i])
            assert_norms_close(norms_of(sd), r["norms"][i], 1e-5, what=f"r{rnd}c{i}")
            w.append(copy.deepcopy(sd))
            assert clients[i].negative == r["neg"][i] and clients[i].active == r["act"][i]
        glob.load_state_dict(R.fedavg(w, [N] * n_cl))
        assert_norms_close(norms_of(glob.state_dict()), r["glob_norms"], 1e-5)
        glob.eval()
        with torch.no_grad():
            _, z = glob(data["image"][:4])
        np.testing.assert_allclose(z.numpy(), np.array(r["probe_logits"]), rtol=1e-4, atol=1e-5)


def test_traj_fixmatch():
    g = load_golden("traj_fixmatch.json")
    C, N = g["C"], g["N"]
    args = make_args(n_classes=C, n_clients=1)
    data = data_dict(N, C, g["hw"], g["data_seed"], True)
    _, neg = class_lists(data["targets"], C)
    net = oracle_net(C, g["init_seed"])
    with torch.no_grad():
        net.fc.weight.mul_(g["fc_scale"])
    cl = R.RefClient(args, 0, data, list(range(N)), neg, [0])
    np.testing.assert_allclose(cl.loss_w_unknown, g["loss_w_unknown"], rtol=0)
    sd, loss, _ = cl.train_fixmatch(net, g["order"])
    assert abs(loss - g["loss"]) <= 1e-5 * abs(g["loss"])
    assert_norms_close(norms_of(sd), g["norms"], 1e-5)


def _perturb_bn(net, seed):
    """the conditioned goldens' initial BatchNorm affine (tests/synth.perturbed_bn, as make_golden.py applies it)"""
    if seed is None:
        return net
    from tests.synth import perturbed_bn
    sd = net.state_dict()
    with torch.no_grad():
        for k, v in perturbed_bn([(k, tuple(t.shape)) for k, t in sd.items()], seed):
            sd[k].copy_(torch.from_numpy(v))
    return net


@pytest.mark.parametrize("name", ["traj_fedmlp", "traj_fedmlp_tail4", "traj_fedmlp_tail1"])
def test_traj_fedmlp_two_stage(name):
    """Full two-stage FedMLP flow incl. prototype pass, tagging, selection, FedAvg*.  The tail goldens (N = 100 / 97 at
    bs 32) end every pass with a batch of 4 / of ONE image per view, normalised by args.batch_size
    (utils/local_training.py:47-48, 956-959)."""
    g = load_golden(name + ".json")
    P = np.load(GOLDEN + f"/{name}_protos.npz")
    C, n_cl, N, S1 = g["C"], g["n_clients"], g["N"], g["S1"]
    args = make_args(n_classes=C, n_clients=n_cl, rounds_FedMLP_stage1=S1,
                     clean_threshold=g.get("clean_threshold", 0.005), noise_threshold=g.get("noise_threshold", 0.01))
    data = data_dict(n_cl * N, C, g["hw"], g["data_seed"], True)
    _, neg = class_lists(data["targets"], C)
    glob = _perturb_bn(oracle_net(C, g["init_seed"]), g.get("bn_seed"))
    clients = [R.RefClient(args, i, data, g["users"][i], neg, [i]) for i in range(n_cl)]
    assert [c.negative for c in clients] == g["neg_lists"]
    prototype, lens = None, [N] * n_cl
    for rnd, r in enumerate(g["rounds"]):
        w, taos, protos = [], [], []
        for i, cl in enumerate(clients):
            net = copy.deepcopy(glob)
            if rnd < S1:
                ret = cl.stage1(net, r["train_orders"][i], with_proto=(rnd == S1 - 1),
                                negative_param=g["neg_lists"][i])
            else:
                ret = cl.stage2(rnd, net, prototype, g["neg_lists"][i], r["feat_orders"][i],
                                r["train_orders"][i])
                assert cl.traindata_idx == r["traindata_idx"][i]
                assert cl.class_num_list == r["class_num_list"][i]
            assert abs(ret[1] - r["loss"][i]) <= 1e-5 * abs(r["loss"][i])
            assert_norms_close(norms_of(ret[0]), r["norms"][i], 1e-5, what=f"r{rnd}c{i}")
            w.append(copy.deepcopy(ret[0]))
            if len(ret) >= 5:
                taos.append(ret[3]); protos.append(ret[4])
                np.testing.assert_array_equal(ret[3], P[f"r{rnd}_c{i}_t"])
                np.testing.assert_allclose(ret[4].numpy(), P[f"r{rnd}_c{i}_proto"],
                                           rtol=1e-5, atol=1e-6)
        glob.load_state_dict(R.fedavg(w, lens))
        if rnd >= S1 - 1:
            tao = R.fedavg_tao(taos, lens, g["class_negative_client_list"])
            prototype = R.fedavg_proto(protos, lens, g["class_active_client_list"])
            np.testing.assert_allclose(tao, r["tao"], rtol=1e-12)
            np.testing.assert_allclose(prototype.numpy(), P[f"r{rnd}_glob_proto"],
                                       rtol=1e-5, atol=1e-6, equal_nan=True)
        assert_norms_close(norms_of(glob.state_dict()), r["glob_norms"], 1e-5)


@pytest.mark.parametrize("tail", [4, 1])
def test_traj_fixmatch_tails(tail):
    """train_FixMatch with a last batch of 4 / of ONE image (two views), bs_norm = 32"""
    g = load_golden("traj_fixmatch_tails.json")[f"tail{tail}"]
    C, N = g["C"], g["N"]
    args = make_args(n_classes=C, n_clients=1)
    data = data_dict(N, C, g["hw"], g["data_seed"], True)
    _, neg = class_lists(data["targets"], C)
    net = _perturb_bn(oracle_net(C, g["init_seed"]), g["bn_seed"])
    with torch.no_grad():
        net.fc.weight.mul_(g["fc_scale"])
    cl = R.RefClient(args, 0, data, list(range(N)), neg, [0])
    np.testing.assert_allclose(cl.loss_w_unknown, g["loss_w_unknown"], rtol=0)
    sd, loss, _ = cl.train_fixmatch(net, g["order"])
    assert abs(loss - g["loss"]) <= 1e-5 * abs(g["loss"])
    assert_norms_close(norms_of(sd), g["norms"], 1e-5)


def test_step224():
    g = load_golden("step224.json")
    C, N = g["C"], g["N"]
    args = make_args(n_classes=C, n_clients=1, batch_size=g["bs"])
    net0 = oracle_net(C, g["init_seed"])
    d2 = data_dict(N, C, g["hw"], g["data_seed"], True)
    d1 = data_dict(N, C, g["hw"], g["data_seed"], False)
    _, neg = class_lists(d1["targets"], C)
    net0.eval()
    with torch.no_grad():
        _, z = net0(d2["image_aug_1"][:4])
    np.testing.assert_allclose(z.numpy(), np.array(g["init_probe_logits"]), rtol=1e-4, atol=1e-5)
    cl = R.RefClient(args, 0, d1, list(range(N)), neg, [0])
    sd, loss, _ = cl.train(copy.deepcopy(net0), list(range(N)))
    assert abs(loss - g["train"]["loss"]) <= 1e-5 * abs(loss)
    assert_norms_close(norms_of(sd), g["train"]["norms"], 1e-5)
    cl = R.RefClient(args, 0, d2, list(range(N)), neg, [0])
    ret = cl.stage1(copy.deepcopy(net0), list(range(N)), with_proto=False)
    assert abs(ret[1] - g["stage1"]["loss"]) <= 1e-5 * abs(ret[1])
    assert_norms_close(norms_of(ret[0]), g["stage1"]["norms"], 1e-5)


def test_traj_baselines_rscfed_fednoro_cbafed():
    """the oracle's restatement of train_RSCFed / train_FedNoRo (warm-up) /
    train_CBAFed follows the imported reference (tests/golden/traj_baselines.json)."""
    g = load_golden("traj_baselines.json")
    C, N, hw = g["C"], g["N"], g["hw"]
    args = make_args(n_classes=C, n_clients=1)
    # RSCFed
    r = g["rscfed"]
    data = data_dict(N, C, hw, r["data_seed"], True)
    _, neg = class_lists(data["targets"], C)
    cl = R.RefClient(args, 0, data, list(range(N)), neg, [0])
    np.testing.assert_allclose(cl.loss_w, r["loss_w"], rtol=0)
    teacher = oracle_net(C, r["teacher_seed"])
    sd, loss, _ = cl.train_rscfed(oracle_net(C, g["init_seed"]), teacher, r["order"])
    assert abs(loss - r["loss"]) <= 1e-5 * abs(r["loss"])
    assert_norms_close(norms_of(sd), r["norms"], 1e-5)
    assert_norms_close(norms_of(teacher.state_dict()), r["teacher_norms"], 1e-6)
    assert cl.negative == r["neg"] and cl.active == r["act"]
    # FedNoRo warm-up
    r = g["fednoro"]
    data = data_dict(N, C, hw, r["data_seed"], False)
    _, neg = class_lists(data["targets"], C)
    cl = R.RefClient(args, 0, data, list(range(N)), neg, [1])
    w_kd = R.consistency_weight(r["rnd"], r["begin"], r["end"]) * r["a"]
    assert abs(w_kd - r["weight_kd"]) < 1e-12
    sd, loss, _ = cl.train_fednoro(oracle_net(C, g["init_seed"]), r["order"], w_kd)
    assert abs(loss - r["loss"]) <= 1e-5 * abs(r["loss"])
    assert_norms_close(norms_of(sd), r["norms"], 1e-5)
    np.testing.assert_allclose(cl.class_num_list, r["class_num_list"], rtol=0)
    # CBAFed: warm-up round then pseudo-labelling round
    r = g["cbafed"]
    data = data_dict(N, C, hw, r["data_seed"], False)
    _, neg = class_lists(data["targets"], C)
    cl = R.RefClient(args, 0, data, list(range(N)), neg, [2])
    net = oracle_net(C, g["init_seed"])
    sd, loss, _, cnum, dnum = cl.train_cbafed(net, r["orders"][0])
    assert abs(loss - r["loss"][0]) <= 1e-5 * abs(r["loss"][0])
    assert_norms_close(norms_of(sd), r["norms"][0], 1e-5)
    assert cnum == r["class_num_list"][0] and dnum == r["data_num"][0]
    sd, loss, _, cnum, dnum = cl.train_cbafed(net, r["orders"][1], tao=r["tao"])
    assert abs(loss - r["loss"][1]) <= 1e-5 * abs(r["loss"][1])
    assert_norms_close(norms_of(sd), r["norms"][1], 1e-5)
    assert cnum == r["class_num_list"][1] and dnum == r["data_num"][1]
    np.testing.assert_allclose(cl.loss_w, r["loss_w_after"], rtol=1e-12)


def test_augment_oracle_is_bit_exact_with_pillow_fixture():
    """oracle/augment_ref.py (fixed-point nearest affine + flip + ToTensor + Normalize) against outputs of Pillow
    itself recorded by tests/golden/make_augment_golden.py: uint8 and float32 bit for bit."""
    import os
    from oracle.augment_ref import affine_nearest_u8, augment_ref
    from fedmlp_amd.augment import IMAGENET_MEAN, IMAGENET_STD
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "augment_pil.npz"))
    for i in range(len(g["images"])):
        a = affine_nearest_u8(g["images"][i], g["matrices"][i])
        if g["flips"][i]:
            a = a[:, :, ::-1]
        np.testing.assert_array_equal(a, g["out_u8"][i])
        np.testing.assert_array_equal(augment_ref(g["images"][i], g["matrices"][i], g["flips"][i], IMAGENET_MEAN,
                                                  IMAGENET_STD), g["out_f32"][i])
