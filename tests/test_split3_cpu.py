"""The arithmetic behind the conv GEMMs' split-product form (fedmlp_amd/csrc/split3.h), restated in numpy: the three-way bf16
split of an fp32 value is exact, the nine partial products are exact and sum to the fp32 product, and the three that the
shipped six-product form leaves out are at most 2^-24 of it each.  (The kernels themselves are checked against float64 convolutions on
the GPU: tests/test_kernels_gpu.py::test_split_products_are_fp32_accurate.)"""
import numpy as np


def bf16_rne(x):
    """round an fp32 array to bf16 (nearest even), returned as fp32 -- what v_cvt_pk_bf16_f32 does for finite values"""
    u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def split3(x):
    h = bf16_rne(x)
    r = (x - h).astype(np.float32)
    m = bf16_rne(r)
    l = (r - m).astype(np.float32)
    return h, r, m, l


def _values(n, seed, emax=40):
    rs = np.random.RandomState(seed)
    x = (rs.standard_normal(n) * np.exp2(rs.randint(-emax, emax, n))).astype(np.float32)
    edge = [0.0, -0.0, 1.0, -1.0, 3.0, 1.0 + 2.0 ** -23, 2.0 - 2.0 ** -23, 1.0 + 2.0 ** -8, 1.0 + 2.0 ** -9,
            1.0 + 3 * 2.0 ** -9, 255.0, 256.0, 257.0, 0.1] + ([1e-30, 1e30] if emax >= 40 else [2.0 ** -20, 2.0 ** 20])
    x[:4 * len(edge)] = np.array(edge * 4, np.float32)
    return x


def test_three_way_split_is_exact():
    x = _values(1 << 20, 1)
    h, r, m, l = split3(x)
    x64 = x.astype(np.float64)
    assert np.array_equal(x64 - h.astype(np.float64), r.astype(np.float64))          # x - h is exact in fp32
    assert np.array_equal(r.astype(np.float64) - m.astype(np.float64), l.astype(np.float64))
    assert np.array_equal(bf16_rne(l), l)                                              # l IS a bf16: the third plane loses nothing
    assert np.array_equal(h.astype(np.float64) + m.astype(np.float64) + l.astype(np.float64), x64)
    ax = np.abs(x64)
    assert (np.abs(m) <= 2.0 ** -8 * ax).all() and (np.abs(l) <= 2.0 ** -16 * ax).all()


def test_nine_partial_products_are_the_product_and_six_are_within_2_pow_minus_26():
    a, b = _values(1 << 18, 2, 20), _values(1 << 18, 3, 20)       # (products of the smallest planes stay inside fp32's normal range)
    ah, _, am, al = split3(a)
    bh, _, bm, bl = split3(b)
    f = lambda v: v.astype(np.float64)
    exact = f(a) * f(b)                                                                # 24 x 24 bits: exact in float64
    parts = {(i, j): f(p) * f(q) for i, p in enumerate((ah, am, al)) for j, q in enumerate((bh, bm, bl))}
    for v in parts.values():                                                           # 8 x 8 significant bits: each is exact in fp32
        assert np.array_equal(v.astype(np.float32).astype(np.float64), v)
    nine = sum(parts.values())
    assert np.array_equal(nine, exact)
    six = nine - parts[(1, 2)] - parts[(2, 1)] - parts[(2, 2)]
    # |m| <= 2^-8 |x|, |l| <= 2^-16 |x|: the three terms are at most 2^-24, 2^-24, 2^-32 of the product -- two unit roundoffs
    # (2^-24) at worst, against the K of them that the fp32 accumulation of a K-term dot product carries (K = 64 ... 4608 here)
    assert (np.abs(six - exact) <= 2.0 ** -23 * np.abs(exact)).all()
    ok = np.abs(exact) > 0
    worst = np.abs((six - exact)[ok] / exact[ok]).max()
    assert 2.0 ** -26 < worst <= 2.0 ** -24, worst                                      # measured 2^-24.3
