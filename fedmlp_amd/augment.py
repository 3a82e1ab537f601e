"""Host side of the HBM-resident input pipeline.

The reference's train transform (dataset/dataset.py:40-53) is
Resize(224) -> RandomAffine(degrees=10, translate=(0.02, 0.02)) -> RandomHorizontalFlip ->
ToTensor -> Normalize(ImageNet mean/std), applied on PIL images by DataLoader workers, once per
view and per __getitem__ (dataset/all_dataset.py:66-91).  Resize precedes every random op, so
caching the resized uint8 pixels in HBM is exact; the random draws happen here (torchvision 0.13's
RandomAffine.get_params / RandomHorizontalFlip semantics and its _get_inverse_affine_matrix, restated:
torchvision is not vendored in the reference), the pixel work is the engine's fm_augment kernel, which
is bit-exact with Pillow's fixed-point nearest-neighbour affine (checked against Pillow's own outputs,
tests/golden/augment_pil.npz).  The ORDER of the reference's draws (worker-seeded RNG streams) is not
reproducible by construction: "parity unpinned" for that part only.
"""
import math

import numpy as np
import torch

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def inverse_affine_matrix(center, angle, translate):
    """torchvision.transforms.functional._get_inverse_affine_matrix with scale 1, shear 0."""
    rot = math.radians(angle)
    cx, cy = center
    tx, ty = translate
    a, b, c, d = math.cos(rot), -math.sin(rot), math.sin(rot), math.cos(rot)
    m = [d, -b, 0.0, -c, a, 0.0]
    m[2] += m[0] * (-cx - tx) + m[1] * (-cy - ty)
    m[5] += m[3] * (-cx - tx) + m[4] * (-cy - ty)
    m[2] += cx
    m[5] += cy
    return m


def _fix16(v):
    t = float(v) * 65536.0 + 0.5                # Geometry.c FIX(): FLOOR(v * 65536.0 + 0.5)
    return int(t) if t >= 0.0 else int(math.floor(t))


def fixed_point_params(matrix, flip):
    """6 doubles + flip -> the int32[8] record fm_augment takes (include/fedmlp_hip.h)."""
    a0, a1, a2, a3, a4, a5 = [float(v) for v in matrix[:6]]
    return [_fix16(a0), _fix16(a1), _fix16(a2 + a0 * 0.5 + a1 * 0.5), _fix16(a3), _fix16(a4),
            _fix16(a5 + a3 * 0.5 + a4 * 0.5), int(bool(flip)), 0]


def draw_matrices(B, H, W, generator=None, degrees=10.0, translate=(0.02, 0.02), p_flip=0.5):
    """One RandomAffine + RandomHorizontalFlip draw per sample: (matrices [B,6] float64, flips [B]).
    angle ~ U(-deg, deg); tx, ty = round(U(-t*W, t*W)), round(U(-t*H, t*H)); flip ~ U(0,1) < p."""
    u = torch.rand((B, 4), generator=generator).numpy().astype(np.float64)
    mats = np.zeros((B, 6), np.float64)
    flips = np.zeros(B, np.int32)
    for b in range(B):
        angle = -degrees + 2 * degrees * u[b, 0]
        tx = int(round(-translate[0] * W + 2 * translate[0] * W * u[b, 1]))
        ty = int(round(-translate[1] * H + 2 * translate[1] * H * u[b, 2]))
        mats[b] = inverse_affine_matrix((W * 0.5, H * 0.5), angle, (tx, ty))
        flips[b] = 1 if u[b, 3] < p_flip else 0
    return mats, flips


def draw_params(B, H, W, generator=None, **kw):
    """[B,8] int32 parameter records of one draw per sample."""
    mats, flips = draw_matrices(B, H, W, generator, **kw)
    return np.asarray([fixed_point_params(mats[b], flips[b]) for b in range(B)], dtype=np.int32)


def identity_params(B):
    """the test-time transform (dataset/dataset.py:55-60: Resize -> ToTensor -> Normalize): no affine, no flip"""
    return np.asarray([fixed_point_params([1.0, 0.0, 0.0, 0.0, 1.0, 0.0], 0)] * B, dtype=np.int32)


class CachedAugmentedViews:
    """Per-client uint8 cache in HBM (N x 3 x H x W bytes: 752 MB for 5 000 ICH images) that hands out
    freshly augmented views of a batch (dataset/all_dataset.py:66-78) without touching the host pixels."""

    def __init__(self, engine, images_u8, mean=IMAGENET_MEAN, std=IMAGENET_STD):
        self.engine = engine
        self.cache = torch.as_tensor(images_u8, dtype=torch.uint8).to(engine.device).contiguous()
        self.mean, self.std = mean, std

    def view(self, sample_idx, params):
        idx = torch.as_tensor(list(sample_idx), dtype=torch.int32, device=self.engine.device)
        p = torch.from_numpy(np.ascontiguousarray(params, dtype=np.int32)).to(self.engine.device)
        return self.engine.augment(self.cache, idx, p, self.mean, self.std)

    def views(self, sample_idx, generator=None, n_views=2):
        H, W = self.engine.in_h, self.engine.in_w
        return [self.view(sample_idx, draw_params(len(sample_idx), H, W, generator)) for _ in range(n_views)]


class AugmentedDataset:
    """dataset/all_dataset.py:64-91 contract over an HBM-resident uint8 cache: every access to "image" /
    "image_aug_1" / "image_aug_2" of the TRAIN set is a fresh RandomAffine + HFlip draw (two independent
    draws for the two views, :75-76); a test set (train=False) gets the deterministic transform.
    LocalUpdate / globaltest ask for whole batches through device_batch(), so the pixels never leave HBM."""

    def __init__(self, images_u8, targets, train=True, generator=None, mean=IMAGENET_MEAN, std=IMAGENET_STD):
        self.images_u8 = torch.as_tensor(images_u8, dtype=torch.uint8)
        self.targets = np.asarray(targets, dtype=np.float32)
        self.train, self.generator = train, generator
        self.mean, self.std = mean, std
        self._cav = None

    def __len__(self):
        return len(self.targets)

    def _host_item(self, i):
        """Host-side single item (shape probing, small tests): same arithmetic on the CPU tensor."""
        x = self.images_u8[i].float().div(255.0)
        m = torch.tensor(self.mean).view(3, 1, 1); s = torch.tensor(self.std).view(3, 1, 1)
        return (x - m) / s

    def __getitem__(self, i):
        x = self._host_item(i)
        return {"image": x, "image_aug_1": x, "image_aug_2": x, "target": self.targets[i].copy(), "index": i}

    def device_batch(self, engine, key, sample_idx):
        if self._cav is None or self._cav.engine is not engine:
            self._cav = CachedAugmentedViews(engine, self.images_u8, self.mean, self.std)
        H, W = engine.in_h, engine.in_w
        p = draw_params(len(sample_idx), H, W, self.generator) if self.train else identity_params(len(sample_idx))
        return self._cav.view(sample_idx, p)
