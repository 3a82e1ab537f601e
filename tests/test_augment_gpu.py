"""HBM-resident input pipeline (dataset/dataset.py:40-53 on a uint8 cache): the fm_augment kernel against
Pillow's own outputs (tests/golden/augment_pil.npz, made by tests/golden/make_augment_golden.py) -- bit-exact,
uint8 pixel choice and fp32 normalisation alike -- and wired into LocalUpdate / globaltest."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "augment_pil.npz")


def test_augment_is_bit_exact_with_pillow_fixture():
    from fedmlp_amd.engine import Engine
    from fedmlp_amd.augment import fixed_point_params, IMAGENET_MEAN, IMAGENET_STD
    g = np.load(GOLD)
    imgs, mats, flips = g["images"], g["matrices"], g["flips"]
    N, _, H, W = imgs.shape
    eng = Engine("Resnet18", 5, H, W, 16)                # 64 x 96: non-square on purpose
    try:
        cache = torch.from_numpy(imgs).to(eng.device)
        params = np.asarray([fixed_point_params(mats[i], flips[i]) for i in range(N)], np.int32)
        idx = torch.arange(N, dtype=torch.int32, device=eng.device)
        out = eng.augment(cache, idx, torch.from_numpy(params).to(eng.device), IMAGENET_MEAN, IMAGENET_STD).cpu().numpy()
        np.testing.assert_array_equal(out, g["out_f32"])         # what Pillow + ToTensor + Normalize produced, bit for bit
    finally:
        eng.close()


def test_augment_flip_and_normalise_bit_exact_vs_oracle_at_224():
    """224x224 (the reference's size), random draws incl. both flip states: kernel == oracle == Pillow semantics."""
    from fedmlp_amd.engine import get_engine
    from fedmlp_amd.augment import draw_matrices, fixed_point_params, IMAGENET_MEAN, IMAGENET_STD
    from oracle.augment_ref import augment_ref
    H = W = 224
    eng = get_engine("Resnet18", 5, H, W, 8)
    rs = np.random.RandomState(0)
    imgs = rs.randint(0, 256, size=(6, 3, H, W)).astype(np.uint8)
    sel = [5, 0, 3, 3, 1]
    mats, flips = draw_matrices(len(sel), H, W, torch.Generator().manual_seed(1))
    flips[0], flips[1] = 1, 0
    params = np.asarray([fixed_point_params(mats[b], flips[b]) for b in range(len(sel))], np.int32)
    out = eng.augment(torch.from_numpy(imgs).to(eng.device), torch.as_tensor(sel, dtype=torch.int32, device=eng.device),
                      torch.from_numpy(params).to(eng.device), IMAGENET_MEAN, IMAGENET_STD).cpu().numpy()
    for b, s in enumerate(sel):
        np.testing.assert_array_equal(out[b], augment_ref(imgs[s], mats[b], flips[b], IMAGENET_MEAN, IMAGENET_STD))


def test_augmentation_is_in_the_training_path():
    """LocalUpdate on an AugmentedDataset: batches come from the uint8 HBM cache through fm_augment (fresh draws per
    view), the test-time transform is the deterministic one, and a FedMLP stage-1 round runs on it."""
    from fedmlp_amd.augment import AugmentedDataset, IMAGENET_MEAN, IMAGENET_STD
    from fedmlp_amd.model import build_model
    from fedmlp_amd.local_training import LocalUpdate
    from fedmlp_amd.evaluations import globaltest
    from tests.helpers import make_args
    from tests.synth import class_lists
    C, HW, N = 4, 64, 48
    rs = np.random.RandomState(3)
    imgs = rs.randint(0, 256, size=(N, 3, HW, HW)).astype(np.uint8)
    targets = (rs.uniform(size=(N, C)) < 0.3).astype(np.float32)
    targets[0, :] = 1                                   # every class has a positive
    ds = AugmentedDataset(imgs, targets, train=True, generator=torch.Generator().manual_seed(5))
    args = make_args(n_classes=C, batch_size=16, rounds_FedMLP_stage1=2)
    pos, neg = class_lists(targets, C)
    net = build_model(make_args(n_classes=C, pretrained=0, batch_size=16))
    loc = LocalUpdate(args, 1, ds, list(range(N)), pos, neg, active_class_list=[1])
    eng = loc._bind(net, "image_aug_1")
    v1 = loc._images(eng, "image_aug_1", [0, 1, 2])
    v2 = loc._images(eng, "image_aug_2", [0, 1, 2])
    assert v1.shape == (3, 3, HW, HW) and v1.is_cuda and not torch.equal(v1, v2)     # two independent draws
    plain = torch.stack([ds._host_item(i) for i in (0, 1, 2)])
    assert not torch.equal(v1.cpu(), plain)                                             # really transformed
    ret = loc.train_FedMLP(0, [0] * C, None, None, None, None, net=net)
    assert np.isfinite(ret[1])
    test = AugmentedDataset(imgs[:16], targets[:16], train=False)
    tb = test.device_batch(eng, "image", [3, 4])
    np.testing.assert_array_equal(tb.cpu().numpy(), torch.stack([test._host_item(3), test._host_item(4)]).numpy())
    m = globaltest(net, test, make_args(n_classes=C, batch_size=4))
    assert 0.0 <= float(m["mAP"]) <= 1.0
