"""Synthetic data shared by the golden generator and the parity tests.

numpy RandomState (legacy, stream-stable across numpy versions) so the build
container and the GPU box regenerate identical inputs from a seed.
"""
import numpy as np


def synth_arrays(n, C, hw, seed, two_view, p_pos=0.3):
    """-> (targets[n,C] f32 in {0,1}, x1[n,3,hw,hw] f32, x2 or None)."""
    rs = np.random.RandomState(seed)
    targets = (rs.uniform(size=(n, C)) < p_pos).astype(np.float32)
    x1 = rs.standard_normal((n, 3, hw, hw)).astype(np.float32)
    x2 = (x1 + 0.1 * rs.standard_normal((n, 3, hw, hw))).astype(np.float32) if two_view else None
    return targets, x1, x2


def class_lists(targets, C):
    """class_pos_idx / class_neg_idx as main.py:58-66 builds them with p_pos_1 = 0
    (class_neg_idx[c] = every positive of class c, order irrelevant)."""
    pos = [np.where(targets[:, c] == 1)[0] for c in range(C)]
    return pos, [p.copy() for p in pos]


def perturbed_bn(sd_items, seed):
    """Non-trivial BatchNorm affine for the conditioned goldens: walking the state_dict in key order, every BN
    weight becomes 1 + 0.1 n and every BN bias 0.1 n (n ~ N(0,1) from RandomState(seed)).  Yields (key, array)
    only for the entries it replaces.  Used by tests/golden/make_golden.py (on the reference side) and by the GPU
    tests (on the engine side), so both start from bit-identical weights."""
    rs = np.random.RandomState(seed)
    for k, shape in sd_items:
        if len(shape) != 1:
            continue
        is_bn = ("bn" in k) or ("downsample.1" in k)
        if is_bn and k.endswith(".weight"):
            yield k, (1.0 + 0.1 * rs.standard_normal(shape)).astype(np.float32)
        elif is_bn and k.endswith(".bias"):
            yield k, (0.1 * rs.standard_normal(shape)).astype(np.float32)
