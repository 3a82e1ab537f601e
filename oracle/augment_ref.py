"""Oracle for the augmentation kernel (TEST INFRASTRUCTURE ONLY): numpy restatement of
PIL Image.transform(AFFINE, NEAREST, fill 0) -> horizontal flip -> ToTensor -> Normalize, the pixel
work of dataset/dataset.py:40-53.  torchvision / the datasets are absent from the reference tree,
so this piece is "parity unpinned"; it fixes the semantics the kernel is checked against."""
import numpy as np


def augment_ref(img_u8, params, mean, std):
    """img_u8 [3,H,W] uint8, params [8] -> float32 [3,H,W]."""
    _, H, W = img_u8.shape
    m = params[:6].astype(np.float32)
    ys, xs = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    fx = xs.astype(np.float32) + np.float32(0.5)
    fy = ys.astype(np.float32) + np.float32(0.5)
    xin = np.floor(m[0] * fx + m[1] * fy + m[2]).astype(np.int64)
    yin = np.floor(m[3] * fx + m[4] * fy + m[5]).astype(np.int64)
    ok = (xin >= 0) & (xin < W) & (yin >= 0) & (yin < H)
    aff = np.where(ok[None], img_u8[:, np.clip(yin, 0, H - 1), np.clip(xin, 0, W - 1)], 0).astype(np.float32)
    if params[6] != 0:
        aff = aff[:, :, ::-1]
    mean = np.asarray(mean, np.float32)[:, None, None]
    std = np.asarray(std, np.float32)[:, None, None]
    return ((aff / np.float32(255.0) - mean) / std).astype(np.float32)
