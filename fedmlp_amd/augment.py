"""Host side of the HBM-resident input pipeline.

The reference's train transform (dataset/dataset.py:40-53) is
Resize(224) -> RandomAffine(degrees=10, translate=(0.02, 0.02)) -> RandomHorizontalFlip ->
ToTensor -> Normalize(ImageNet mean/std), applied on PIL images by DataLoader workers.
Resize precedes every random op, so caching the resized uint8 pixels in HBM is exact; the
random draws happen here (torchvision's RandomAffine.get_params / RandomHorizontalFlip
semantics restated from torchvision 0.13, which is not vendored in the reference: "parity
unpinned" for the draw order), the pixel work is the engine's fm_augment kernel.
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


def draw_params(B, H, W, generator=None, degrees=10.0, translate=(0.02, 0.02), p_flip=0.5):
    """[B,8] float32: inverse affine m0..m5, flip, 0 -- one RandomAffine + RandomHorizontalFlip draw
    per sample (angle ~ U(-deg, deg); tx, ty ~ round(U(-t*W, t*W)), round(U(-t*H, t*H)))."""
    u = torch.rand((B, 4), generator=generator).numpy().astype(np.float64)
    out = np.zeros((B, 8), np.float32)
    for b in range(B):
        angle = -degrees + 2 * degrees * u[b, 0]
        tx = int(round(-translate[0] * W + 2 * translate[0] * W * u[b, 1]))
        ty = int(round(-translate[1] * H + 2 * translate[1] * H * u[b, 2]))
        out[b, :6] = inverse_affine_matrix((W * 0.5, H * 0.5), angle, (tx, ty))
        out[b, 6] = 1.0 if u[b, 3] < p_flip else 0.0
    return out


class CachedAugmentedViews:
    """Per-client uint8 cache in HBM (N x 3 x H x W bytes: 752 MB for 5 000 ICH images) that hands out
    the two augmented views of a batch (dataset/all_dataset.py:66-78) without touching the host."""

    def __init__(self, engine, images_u8, mean=IMAGENET_MEAN, std=IMAGENET_STD):
        self.engine = engine
        self.cache = torch.as_tensor(images_u8, dtype=torch.uint8).to(engine.device).contiguous()
        self.mean, self.std = mean, std

    def views(self, sample_idx, generator=None, n_views=2):
        H, W = self.engine.in_h, self.engine.in_w
        idx = torch.as_tensor(sample_idx, dtype=torch.int32, device=self.engine.device)
        outs = []
        for _ in range(n_views):
            p = torch.from_numpy(draw_params(len(sample_idx), H, W, generator)).to(self.engine.device)
            outs.append(self.engine.augment(self.cache, idx, p, self.mean, self.std))
        return outs
