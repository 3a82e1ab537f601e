"""csrc/pwgrad_ring.hip walks PADDED pixel positions (one zero position after each image row, one zero row after each image) and
turns a position into an NHWC pixel index with two multiply-high divisions by ceil(2^32 / d).  This restates that decode in numpy
and checks it against exact integer arithmetic over every position the launcher admits (positions below 2^24, rows of 28 .. 62
pixels), and that every tap of a 3x3 / pad-1 window is the constant shift (kh - 1)(W + 1) + (kw - 1) of the position."""
import numpy as np
import pytest


def pix_of(q, H, W, nimg):
    """the kernel's decode: padded position -> pixel index, -1 = padding"""
    Wp, Hp = W + 1, H + 1
    Z0 = Wp + 1
    magW = (1 << 32) // Wp + 1
    magH = (1 << 32) // Hp + 1
    t = q.astype(np.int64) - Z0
    tu = t & 0xFFFFFFFF                                   # (unsigned)t
    row = (tu * magW) >> 32                               # __umulhi
    col = (tu - row * Wp) & 0xFFFFFFFF
    img = ((row & 0xFFFFFFFF) * magH) >> 32
    oh = (row - img * Hp) & 0xFFFFFFFF
    ok = (t >= 0) & (col < W) & (oh < H) & (img < nimg)
    return np.where(ok, (img * H + oh) * W + col, -1)


@pytest.mark.parametrize("H,W,nimg", [(56, 56, 512), (28, 28, 2048), (62, 62, 300), (31, 29, 1000), (56, 56, 3)])
def test_decode_matches_exact_arithmetic(H, W, nimg):
    Wp, Hp = W + 1, H + 1
    Z0 = Wp + 1
    Q = Z0 + nimg * Hp * Wp
    assert Q + 64 < (1 << 24)                          # the launcher's admission bound (pwgrad_ring_takes)
    rng = np.random.default_rng(7)
    q = np.unique(np.concatenate([np.arange(-70, 4000), np.arange(Q - 4000, Q + 300),
                                  rng.integers(0, Q, 200000)])).astype(np.int64)
    got = pix_of(q, H, W, nimg)
    t = q - Z0
    row, col = np.divmod(t, Wp)
    img, oh = np.divmod(row, Hp)
    ok = (t >= 0) & (col < W) & (oh < H) & (img < nimg)
    want = np.where(ok, (img * H + oh) * W + col, -1)
    np.testing.assert_array_equal(got, want)
    # every real pixel is reached exactly once
    allq = np.arange(0, Q, dtype=np.int64)
    px = pix_of(allq, H, W, nimg) if Q < 3_000_000 else None
    if px is not None:
        real = px[px >= 0]
        assert real.size == nimg * H * W and np.array_equal(np.sort(real), np.arange(nimg * H * W))


def test_taps_are_constant_shifts():
    H, W, nimg = 28, 30, 3
    Wp, Hp = W + 1, H + 1
    Z0 = Wp + 1
    img, oh, ow = np.meshgrid(np.arange(nimg), np.arange(H), np.arange(W), indexing="ij")
    q = Z0 + (img * Hp + oh) * Wp + ow                   # position of pixel (img, oh, ow)
    for kh in range(3):
        for kw in range(3):
            src = pix_of((q + (kh - 1) * Wp + (kw - 1)).ravel(), H, W, nimg).reshape(q.shape)
            ih, iw = oh + kh - 1, ow + kw - 1
            inside = (ih >= 0) & (ih < H) & (iw >= 0) & (iw < W)
            want = np.where(inside, (img * H + ih) * W + iw, -1)
            np.testing.assert_array_equal(src, want)


def test_pconv_reciprocal_divmod_is_exact():
    """csrc/pconv.hip decodes pixel indices below 2^22 with q = (int)((float)x * (1.0f / d)) and ONE correction step either way
    (pc_divmod); the float estimate is never off by more than one for the divisors the engine meets (image sizes 7 .. 224 and
    their squares)."""
    rng = np.random.default_rng(11)
    for d in [7, 14, 28, 49, 56, 112, 196, 224, 784, 3136, 12544, 50176, 3, 5, 31, 57 * 57]:
        x = np.unique(np.concatenate([np.arange(0, 70000), rng.integers(0, 1 << 22, 300000),
                                      np.arange((1 << 22) - 70000, 1 << 22)])).astype(np.int64)
        rcp = np.float32(1.0) / np.float32(d)
        q = (x.astype(np.float32) * rcp).astype(np.int64)           # truncation, as the C cast
        r = x - q * d
        lo, hi = r < 0, r >= d
        q = np.where(lo, q - 1, np.where(hi, q + 1, q))
        r = np.where(lo, r + d, np.where(hi, r - d, r))
        np.testing.assert_array_equal(q, x // d)
        np.testing.assert_array_equal(r, x % d)
