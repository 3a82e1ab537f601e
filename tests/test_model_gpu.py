"""Model-object behaviour on the GPU: engine-cache replacement keeps trained state,
ImageNet-style checkpoints load into a C-class net and train (model/all_models.py:99-130)."""
import copy

import numpy as np
import pytest
import torch

from fedmlp_amd import spec
from tests.helpers import make_args

pytestmark = pytest.mark.gpu

HW = 64


def _x(B, seed):
    return torch.randn((B, 3, HW, HW), generator=torch.Generator().manual_seed(seed))


def test_larger_batch_after_training_keeps_the_trained_state():
    """get_engine replaces the cached engine when a larger max_images is asked for.  The resident net's
    trained, device-only state must be pulled first: train a step at a small batch, then call net(x) with
    a batch larger than the engine was built for, and compare with the state read before the growth."""
    from fedmlp_amd.model import build_model
    from fedmlp_amd.engine import get_engine
    C = 3                                       # a class count no other test uses: a fresh cache entry
    net = build_model(make_args(n_classes=C, pretrained=0, batch_size=2))
    eng = net.bind(HW, HW, 8)
    assert eng.max_images == 8
    eng.adam_reset(1e-3)
    lo = torch.zeros(1, device=eng.device)
    y = torch.zeros((4, C), device=eng.device); y[0, 1] = 1
    eng.step_bce(_x(4, 1).to(eng.device), y, [1.0] * C, 4, lo)
    net.mark_trained()
    trained, _ = eng.get_state()
    init, _ = spec.init_state("Resnet18", C, 1037)
    assert np.abs(trained - init).max() > 1e-4   # the step really moved the weights
    net.eval()
    f, z = net(_x(24, 2))                        # 24 > 8: the engine is rebuilt
    eng2 = get_engine("Resnet18", C, HW, HW, 1)
    assert eng2 is not eng and eng2.max_images >= 24 and eng.h is None
    sd = net.state_dict()
    flat, _ = spec.state_dict_to_flat("Resnet18", C, sd)
    np.testing.assert_array_equal(flat, trained)
    # and the forward on the rebuilt engine used the trained weights
    eng2.set_state(trained, np.zeros(eng2.ni, np.int64) + 1)
    f2, z2 = eng2.forward_eval(_x(24, 2).to(eng2.device))
    torch.testing.assert_close(z, z2, rtol=0, atol=0)


def test_two_local_updates_with_different_batch_sizes_share_one_net():
    from fedmlp_amd.model import build_model
    from fedmlp_amd.local_training import LocalUpdate
    from tests.test_local_training_gpu import SynthDataset
    from tests.synth import class_lists
    C = 4
    ds = SynthDataset(24, C, HW, 5, False)
    pos, neg = class_lists(ds.targets, C)
    net = build_model(make_args(n_classes=C, pretrained=0, batch_size=4))
    small = LocalUpdate(make_args(n_classes=C, batch_size=4), 0, ds, list(range(24)), pos, neg, active_class_list=[0])
    big = LocalUpdate(make_args(n_classes=C, batch_size=12), 1, ds, list(range(24)), pos, neg, active_class_list=[1])
    sd1 = small.train(0, net, None)[0]
    sd2 = big.train(1, net, None)[0]             # 4*12 > 4*4 images: the engine grows under the trained net
    assert not torch.equal(sd1["conv1.weight"], sd2["conv1.weight"])
    for k, v in sd2.items():
        assert torch.isfinite(v.float()).all(), k
    # the second round started from the first round's result, not from the initial weights
    assert int(sd2["bn1.num_batches_tracked"]) == int(sd1["bn1.num_batches_tracked"]) + 2


@pytest.mark.parametrize("model", ["Resnet18", "Efficient_b0"])
def test_pretrained_style_checkpoint_loads_and_trains(model, tmp_path, monkeypatch):
    """A 1000-class checkpoint with the package's key names (what `pretrained=True` downloads in the
    reference) loads into a C = 5 net through build_model(args.pretrained=1): backbone taken, classifier
    fresh; the net then runs a training step and an eval forward that agree with the oracle model loaded
    the same way."""
    from fedmlp_amd.model import build_model, PRETRAINED_FILES
    C = 5
    flat, cnt = spec.init_state(model, 1000, 77)
    sd1000 = {k: torch.from_numpy(np.asarray(v)) for k, v in spec.flat_to_state_dict(model, 1000, flat, cnt).items()}
    torch.save(sd1000, tmp_path / PRETRAINED_FILES[model])
    monkeypatch.setenv("FEDMLP_PRETRAINED_DIR", str(tmp_path))
    net = build_model(make_args(model=model, n_classes=C, pretrained=1, batch_size=2))
    sd = net.state_dict()
    ck = spec.classifier_keys(model)
    assert sd[ck[0]].shape[0] == C
    some = "layer2.0.conv1.weight" if model == "Resnet18" else "_blocks.3._project_conv.weight"
    assert torch.equal(sd[some], sd1000[some])
    if model == "Resnet18":
        from oracle.resnet18_ref import ResNet18Ref
        ref = ResNet18Ref(C)
    else:
        from oracle.efficientnet_ref import EfficientNetB0Ref
        ref = EfficientNetB0Ref(C)
    ref.load_state_dict(sd)
    ref.eval()
    x = _x(3, 4)
    with torch.no_grad():
        _, zr = ref(x)
    net.eval()
    _, z = net(x)
    np.testing.assert_allclose(z.cpu().numpy(), zr.numpy(), rtol=2e-4, atol=2e-5)
    eng = net.bind(HW, HW, 8)
    eng.stochastic = False
    eng.adam_reset(3e-5)
    lo = torch.zeros(1, device=eng.device)
    y = torch.zeros((3, C), device=eng.device); y[1, 2] = 1
    eng.step_bce(x.to(eng.device), y, [1.0] * C, 3, lo)
    assert np.isfinite(lo.item())
