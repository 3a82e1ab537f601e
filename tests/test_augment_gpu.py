"""HBM-resident input pipeline: the augmentation kernel vs its numpy oracle on seeded draws."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_augment_matches_oracle():
    from fedmlp_amd.engine import get_engine
    from fedmlp_amd.augment import CachedAugmentedViews, draw_params, IMAGENET_MEAN, IMAGENET_STD
    from oracle.augment_ref import augment_ref
    H = W = 64
    eng = get_engine("Resnet18", 5, H, W, 16)
    rs = np.random.RandomState(0)
    imgs = rs.randint(0, 256, size=(10, 3, H, W)).astype(np.uint8)
    cav = CachedAugmentedViews(eng, imgs)
    g = torch.Generator().manual_seed(1)
    sel = [7, 0, 3, 3, 9]
    params = draw_params(len(sel), H, W, g)
    params[0, 6], params[1, 6] = 1.0, 0.0                 # make sure both flip states are covered
    idx = torch.as_tensor(sel, dtype=torch.int32, device=eng.device)
    out = eng.augment(cav.cache, idx, torch.from_numpy(params).to(eng.device), IMAGENET_MEAN, IMAGENET_STD)
    got = out.cpu().numpy()
    nbad = 0
    for b, s in enumerate(sel):
        want = augment_ref(imgs[s], params[b], IMAGENET_MEAN, IMAGENET_STD)
        # the source-pixel choice is a floor() of an fp32 expression: allow a few boundary pixels
        # (GPU fma contraction vs numpy) but every other pixel must agree to rounding
        diff = np.abs(got[b] - want) > 1e-5
        nbad += int(diff.any(axis=0).sum())
    assert nbad <= 8, nbad
    v1, v2 = cav.views(sel, g)
    assert v1.shape == (len(sel), 3, H, W) and not torch.equal(v1, v2)
