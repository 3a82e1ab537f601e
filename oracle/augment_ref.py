"""Oracle for the augmentation kernel (TEST INFRASTRUCTURE ONLY).

Restates the pixel work of the reference's train transform (dataset/dataset.py:40-53:
RandomAffine(degrees=10, translate=(0.02, 0.02)) -> RandomHorizontalFlip -> ToTensor -> Normalize) as
torchvision 0.13 executes it on PIL images: Image.transform(size, AFFINE, inverse matrix, NEAREST,
fillcolor=0), Image.transpose(FLIP_LEFT_RIGHT), uint8 -> float32 / 255, (v - mean) / std.

Pillow's nearest-neighbour affine runs in 16.16 FIXED POINT (libImaging/Geometry.c affine_fixed): the six
coefficients are rounded to 1/65536, the source coordinate of output pixel (x, y) is
    xin = (FIX(a2 + a0/2 + a1/2) + a0' x + a1' y) >> 16,   yin likewise with a3', a4', a5
and pixels whose source falls outside the image keep the fill value.  Integer arithmetic, so the HIP
kernel can be (and is) bit-exact with it.

Pinned: tests/golden/augment_pil.npz holds outputs of Pillow itself (tests/golden/make_augment_golden.py,
run in the build container where Pillow 12.2 is installed); tests/test_oracle_golden.py checks this file
against them exactly.  Not pinned: the ORDER in which the reference's DataLoader workers draw the random
parameters (worker-seeded torch RNG), which no deterministic restatement can reproduce.
"""
import math

import numpy as np


def fix16(v):
    """Geometry.c: #define FIX(v) FLOOR((v) * 65536.0 + 0.5), FLOOR(v) = v >= 0 ? (int)v : (int)floor(v)"""
    t = float(v) * 65536.0 + 0.5
    return int(t) if t >= 0.0 else int(math.floor(t))


def fixed_coeffs(m):
    """inverse affine matrix (6 doubles, PIL AFFINE convention) -> the six 16.16 integers affine_fixed walks"""
    a0, a1, a2, a3, a4, a5 = [float(v) for v in m[:6]]
    return [fix16(a0), fix16(a1), fix16(a2 + a0 * 0.5 + a1 * 0.5), fix16(a3), fix16(a4), fix16(a5 + a3 * 0.5 + a4 * 0.5)]


def affine_nearest_u8(img_u8, m):
    """img_u8 [3,H,W] uint8 -> Image.transform((W,H), AFFINE, m, NEAREST, fillcolor=0) as uint8 [3,H,W]"""
    _, H, W = img_u8.shape
    c0, c1, c2, c3, c4, c5 = fixed_coeffs(m)
    ys, xs = np.meshgrid(np.arange(H, dtype=np.int64), np.arange(W, dtype=np.int64), indexing="ij")
    xin = (c2 + c0 * xs + c1 * ys) >> 16                 # arithmetic shift = floor, like C's >> on int
    yin = (c5 + c3 * xs + c4 * ys) >> 16
    ok = (xin >= 0) & (xin < W) & (yin >= 0) & (yin < H)
    src = img_u8[:, np.clip(yin, 0, H - 1), np.clip(xin, 0, W - 1)]
    return np.where(ok[None], src, 0).astype(np.uint8)


def augment_ref(img_u8, m, flip, mean, std):
    """[3,H,W] uint8 -> float32 [3,H,W]: affine (NEAREST, fill 0), optional horizontal flip, /255, normalize"""
    a = affine_nearest_u8(img_u8, m)
    if flip:
        a = a[:, :, ::-1]
    t = a.astype(np.float32) / np.float32(255.0)                        # ToTensor: .to(float32).div(255)
    mean = np.asarray(mean, np.float32)[:, None, None]
    std = np.asarray(std, np.float32)[:, None, None]
    return ((t - mean) / std).astype(np.float32)                       # Normalize: sub_(mean).div_(std)
