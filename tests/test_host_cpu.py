"""CPU-only checks of the host side: state layout, the reference-surface helpers
(label masking, pseudo targets, FedAvg* drop-ins) against the reference's golden
vectors, and that the C-ABI library loads and exports every declared symbol."""
import os
import re

import numpy as np
import pytest
import torch

from fedmlp_amd import spec
from tests.helpers import load_golden, make_args
from tests.synth import synth_arrays, class_lists

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_spec_matches_reference_layout():
    nf, ni = spec.sizes("Resnet18", 5)
    assert (nf + ni, ni) == (11188697, 20)            # 122 keys, 11 188 697 elements
    ent = spec.entries("Resnet18", 5)
    assert len(ent) == 122
    from oracle.resnet18_ref import ResNet18Ref
    ref = ResNet18Ref(5).state_dict()
    assert [k for k, _, _ in ent] == list(ref.keys())
    for k, shape, _ in ent:
        assert tuple(ref[k].shape) == tuple(shape), k


def test_efficientnet_b0_spec_matches_oracle_module():
    """Efficient_b0 (BASELINE configs 4-5): key order, shapes and the 4,013,953-parameter count
     of fedmlp_amd.spec agree with the oracle module's state_dict; the oracle forward
    returns the (feature, logits) pair the trainer unpacks."""
    import torch
    from oracle.efficientnet_ref import EfficientNetB0Ref, draw_stochastic
    net = EfficientNetB0Ref(5)
    ent = spec.entries("Efficient_b0", 5)
    sd = net.state_dict()
    assert [k for k, _, _ in ent] == list(sd.keys())
    for k, shape, dt in ent:
        assert tuple(sd[k].shape) == tuple(shape), k
        assert (sd[k].dtype == torch.int64) == (dt == "i64"), k
    assert sum(p.numel() for p in net.parameters()) == 4013953
    assert spec.sizes("Efficient_b0", 5) == (4055969, 49)
    flat, cnt = spec.init_state("Efficient_b0", 5, 3)
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in
                         spec.flat_to_state_dict("Efficient_b0", 5, flat, cnt).items()})
    dc, dr = draw_stochastic(2, torch.Generator().manual_seed(0))
    assert dc.shape == (16, 2) and dr.shape == (2, 1280) and float(dc[0].min()) == 1.0
    net.train()
    f, z = net(torch.randn(2, 3, 64, 64), dc, dr)
    assert f.shape == (2, 1280) and z.shape == (2, 5)


def test_state_flat_roundtrip():
    flat, cnt = spec.init_state("Resnet18", 8, 3)
    sd = spec.flat_to_state_dict("Resnet18", 8, flat, cnt)
    f2, c2 = spec.state_dict_to_flat("Resnet18", 8, sd)
    np.testing.assert_array_equal(flat, f2)
    np.testing.assert_array_equal(cnt, c2)
    flat_b, _ = spec.init_state("Resnet18", 8, 3)
    np.testing.assert_array_equal(flat, flat_b)       # deterministic


def test_label_masking_and_pseudo_targets_golden():
    from fedmlp_amd.local_training import mask_targets, pseudo_targets
    g = load_golden("kat.json")["dataset_split"]
    t = np.array(g["targets"], dtype=np.float32)
    loc, ym = mask_targets(t, g["idxs"], g["active"], g["class_neg_idx"])
    np.testing.assert_array_equal(ym, np.array(g["masked"], dtype=np.float32))
    assert loc.astype(np.float64).sum(axis=0).tolist() == g["counts"]
    y, d = pseudo_targets(loc, g["idxs"], g["active"], g["negative"], g["traindata_idx"])
    np.testing.assert_array_equal(y, np.array(g["pseudo_y"], dtype=np.float32))
    np.testing.assert_array_equal(d, np.array(g["pseudo_distill"], dtype=np.float32))


def test_localupdate_ctor_state_golden():
    """LocalUpdate.__init__ (utils/local_training.py:26-55): class counts and pos_weight from
    the UNMASKED labels, checked against the reference-run golden."""
    from fedmlp_amd.local_training import LocalUpdate
    g = load_golden("traj_train.json")
    C, n_cl, N = g["C"], g["n_clients"], g["N"]

    class DS:
        pass
    ds = DS()
    ds.targets, _, _ = synth_arrays(n_cl * N, C, 4, g["data_seed"], False)
    # synth_arrays draws targets first, so any hw regenerates the same labels
    t224, _, _ = synth_arrays(n_cl * N, C, g["hw"], g["data_seed"], False)
    np.testing.assert_array_equal(ds.targets, t224)
    pos, neg = class_lists(ds.targets, C)
    args = make_args(n_classes=C, n_clients=n_cl)
    for i in range(n_cl):
        loc = LocalUpdate(args, i, ds, g["users"][i], pos, neg, active_class_list=[i])
        np.testing.assert_allclose(loc.loss_w, g["loss_w"][i], rtol=0)
        assert loc.negative_class_list == g["rounds"][0]["neg"][i]


def test_fedavg_dropins_golden():
    from fedmlp_amd.fedavg import FedAvg, FedAvg_tao, FedAvg_proto
    kat = load_golden("kat.json")
    g = kat["fedavg"]
    w = [{k: torch.tensor(v, dtype=torch.int64 if "num_batches" in k else torch.float32)
          for k, v in wi.items()} for wi in g["w"]]
    out = FedAvg(w, g["lens"])
    for k, v in g["out"].items():
        assert str(out[k].dtype) == g["out_dtype"][k]
        np.testing.assert_array_equal(out[k].numpy(), np.array(v, dtype=np.float32))
    g = kat["fedavg_tao"]
    np.testing.assert_array_equal(FedAvg_tao([np.array(v) for v in g["t"]], g["weight"], g["clients"]),
                                  np.array(g["out"]))
    g = kat["fedavg_proto"]
    out = FedAvg_proto([torch.tensor(p) for p in g["P"]], g["weight"], g["clients"]).numpy()
    want = np.array([[np.nan if x is None else x for x in r] for r in g["out"]], dtype=np.float32)
    np.testing.assert_array_equal(out, want)


def test_cabi_library_exports_every_declared_symbol():
    """The built .so must load (no GPU needed) and export every function the header declares;
    the ctypes table must cover the header one-to-one."""
    from fedmlp_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "fedmlp_hip.h")).read()
    dbg = open(os.path.join(ROOT, "include", "fedmlp_hip_debug.h")).read()
    assert not re.search(r"\bfm_debug_[a-z0-9_]+\s*\(", hdr), "test hooks belong in include/fedmlp_hip_debug.h"
    declared = set(re.findall(r"\b(fm_[a-z0-9_]+)\s*\(", hdr + dbg))
    declared -= {"fm_engine", "fm_config", "fm_adam"}
    assert declared == set(_lib.SYMBOLS), (declared ^ set(_lib.SYMBOLS))
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("libfedmlp_hip.so not built (run `make`)")
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert b"gfx950" in lib.fm_version()


def test_product_does_not_import_oracle():
    """The shipped package must never route through the oracle (tests/bench only)."""
    pkg = os.path.join(ROOT, "fedmlp_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "oracle" not in src.replace("# oracle", ""), fn


def test_engine_requires_gpu_and_library():
    from fedmlp_amd.engine import Engine
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError):
        Engine("Resnet18", 5, 64, 64, 4)


def test_eval_metrics_golden():
    """numpy AP / ROC-AUC / multilabel metrics vs sklearn + the reference's multilabel_metrixs
    (tests/golden/eval_metrics.json, made by importing utils/evaluations.py)."""
    from fedmlp_amd.evaluations import average_precision, roc_auc, multilabel_metrics
    g = load_golden("eval_metrics.json")["kat"]
    y, p = np.array(g["y"], dtype=np.float32), np.array(g["p"], dtype=np.float32)
    for c in range(4):
        assert abs(average_precision(y[:, c], p[:, c]) - g["AP"][c]) < 1e-12
        assert abs(roc_auc(y[:, c], p[:, c]) - g["AUC"][c]) < 1e-12
    m = multilabel_metrics(y, p)
    for k in ("BACC", "R", "F1", "P", "hamming_loss"):
        assert abs(m[k] - g[k]) < 1e-12, k
    assert abs(float(m["mAP"]) - np.mean(g["AP"])) < 1e-6


def test_checkpoint_roundtrip_reference_format(tmp_path):
    """main.py:237/361 saves torch.save(netglob.state_dict()); a HipNet must write and read that
    file format (122 torchvision keys, OIHW, int64 counters) and an nn.Module must accept it."""
    from fedmlp_amd.model import HipNet
    from oracle.resnet18_ref import ResNet18Ref
    flat, cnt = spec.init_state("Resnet18", 5, 11)
    cnt[:] = 7
    net = HipNet("Resnet18", 5, flat, cnt)
    path = tmp_path / "model_9.pth"
    torch.save(net.state_dict(), path)
    sd = torch.load(path)
    ref = ResNet18Ref(5)
    ref.load_state_dict(sd)                              # strict: same keys, shapes, dtypes
    assert ref.bn1.num_batches_tracked.item() == 7
    net2 = HipNet("Resnet18", 5, np.zeros_like(flat), np.zeros_like(cnt))
    net2.load_state_dict(torch.load(path))
    np.testing.assert_array_equal(net2.flat, flat)
    np.testing.assert_array_equal(net2.counters, cnt)


def test_checkpoint_roundtrip_efficient_b0(tmp_path):
    """The same torch.save(state_dict()) file format for --model Efficient_b0: efficientnet-pytorch key names
    (_conv_stem.weight ... _fc.bias), accepted strictly by the oracle module and read back bit-exactly."""
    from fedmlp_amd.model import HipNet
    from oracle.efficientnet_ref import EfficientNetB0Ref
    flat, cnt = spec.init_state("Efficient_b0", 5, 11)
    cnt[:] = 3
    net = HipNet("Efficient_b0", 5, flat, cnt)
    path = tmp_path / "model_eff.pth"
    torch.save(net.state_dict(), path)
    ref = EfficientNetB0Ref(5)
    ref.load_state_dict(torch.load(path))
    assert ref._bn0.num_batches_tracked.item() == 3
    net2 = HipNet("Efficient_b0", 5, np.zeros_like(flat), np.zeros_like(cnt))
    net2.load_state_dict(torch.load(path))
    np.testing.assert_array_equal(net2.flat, flat)
    np.testing.assert_array_equal(net2.counters, cnt)


# ---- checkpoints: ImageNet-pretrained load with the classifier swap (model/all_models.py:99-130) -------
def _imagenet_like_sd(model, seed=3):
    flat, cnt = spec.init_state(model, 1000, seed)
    sd = spec.flat_to_state_dict(model, 1000, flat, cnt + 5)
    return {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}


@pytest.mark.parametrize("model", ["Resnet18", "Efficient_b0"])
def test_load_state_dict_nonstrict_keeps_classifier_and_takes_backbone(model):
    from fedmlp_amd.model import build_model
    args = make_args(model=model, n_classes=5, pretrained=0)
    net = build_model(args)
    before = {k: v.clone() for k, v in net.state_dict().items()}
    sd = _imagenet_like_sd(model)
    with pytest.raises((RuntimeError, AssertionError)):
        net.load_state_dict(sd)                      # strict: the 1000-class classifier does not fit
    res = net.load_state_dict(sd, strict=False)
    assert res.missing_keys == [] and res.unexpected_keys == []
    assert tuple(res.mismatched_keys) == spec.classifier_keys(model)
    after = net.state_dict()
    for k in after:
        if k in spec.classifier_keys(model):
            assert torch.equal(after[k], before[k]), k          # fresh Linear(D, C) kept (modify_last_layer)
        elif "num_batches_tracked" in k:
            assert int(after[k]) == int(sd[k])
        else:
            assert torch.equal(after[k], sd[k]), k
    # a partial dict (missing keys) leaves the rest alone
    some = {k: sd[k] for k in list(sd)[:7]}
    res = net.load_state_dict(some, strict=False)
    assert len(res.missing_keys) == len(after) - 7


def test_build_model_pretrained_flag(tmp_path, monkeypatch):
    """args.pretrained defaults to 1 in the reference (utils/options.py:26): without a local checkpoint
    build_model must say so loudly; with one it loads the backbone and re-inits the classifier."""
    from fedmlp_amd.model import build_model, PRETRAINED_FILES
    monkeypatch.setenv("TORCH_HOME", str(tmp_path / "nohub"))
    monkeypatch.delenv("FEDMLP_PRETRAINED_DIR", raising=False)
    args = make_args(model="Resnet18", n_classes=5, pretrained=1)
    with pytest.warns(RuntimeWarning, match="no ImageNet checkpoint"):
        scratch = build_model(args)
    sd = _imagenet_like_sd("Resnet18")
    torch.save(sd, tmp_path / PRETRAINED_FILES["Resnet18"])
    monkeypatch.setenv("FEDMLP_PRETRAINED_DIR", str(tmp_path))
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        net = build_model(args)
    got = net.state_dict()
    assert torch.equal(got["layer3.1.conv2.weight"], sd["layer3.1.conv2.weight"])
    assert got["fc.weight"].shape == (5, 512)
    assert torch.equal(got["fc.weight"], scratch.state_dict()["fc.weight"])
    # an explicit path wins
    args.pretrained_path = str(tmp_path / PRETRAINED_FILES["Resnet18"])
    monkeypatch.delenv("FEDMLP_PRETRAINED_DIR")
    assert torch.equal(build_model(args).state_dict()["conv1.weight"], sd["conv1.weight"])
