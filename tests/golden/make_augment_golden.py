#!/usr/bin/env python3
"""Golden vectors for the augmentation path, produced by PILLOW ITSELF in the build container
(Pillow is not a dependency of the product and is absent from the GPU box's test path).

For seeded uint8 images and RandomAffine/HFlip parameters this records what torchvision 0.13's PIL code
path computes (dataset/dataset.py:40-53): Image.transform(size, AFFINE, inverse matrix, NEAREST,
fillcolor=0) -> transpose(FLIP_LEFT_RIGHT) -> ToTensor -> Normalize.  The inverse matrix formula is
torchvision's _get_inverse_affine_matrix (restated in fedmlp_amd/augment.py; torchvision is not vendored).
Writes tests/golden/augment_pil.npz.   usage: python tests/golden/make_augment_golden.py"""
import os
import sys

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from fedmlp_amd.augment import inverse_affine_matrix, IMAGENET_MEAN, IMAGENET_STD   # noqa: E402


def main():
    rs = np.random.RandomState(20240)
    H, W, N = 64, 96, 6            # multiples of 32 (the engine's input-size granularity), non-square on purpose
    imgs = rs.randint(0, 256, size=(N, 3, H, W)).astype(np.uint8)
    mats, flips, out_u8, out_f32 = [], [], [], []
    for i in range(N):
        angle = float(rs.uniform(-10, 10)) if i != 3 else 0.0       # 0.0: Pillow's scale-only path
        tx = int(round(rs.uniform(-0.02 * W, 0.02 * W)))
        ty = int(round(rs.uniform(-0.02 * H, 0.02 * H)))
        if i == 5:
            angle, tx, ty = 10.0, 1, -1                                       # the extreme of the reference's range
        m = inverse_affine_matrix((W * 0.5, H * 0.5), angle, (tx, ty))
        flip = int(i % 2)
        pil = Image.fromarray(imgs[i].transpose(1, 2, 0)).transform((W, H), Image.AFFINE, m, Image.NEAREST, fillcolor=0)
        if flip:
            pil = pil.transpose(Image.FLIP_LEFT_RIGHT)
        a = np.array(pil).transpose(2, 0, 1)
        t = a.astype(np.float32) / np.float32(255.0)
        t = (t - np.asarray(IMAGENET_MEAN, np.float32)[:, None, None]) / np.asarray(IMAGENET_STD, np.float32)[:, None, None]
        mats.append(m); flips.append(flip); out_u8.append(a); out_f32.append(t.astype(np.float32))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "augment_pil.npz"), images=imgs,
                        matrices=np.asarray(mats, np.float64), flips=np.asarray(flips, np.int32),
                        out_u8=np.asarray(out_u8), out_f32=np.asarray(out_f32))
    print("wrote augment_pil.npz", imgs.shape)


if __name__ == "__main__":
    main()
