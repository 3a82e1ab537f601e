"""Shared helpers for the oracle / engine parity tests."""
import json
import os
import types

import numpy as np
import torch

from fedmlp_amd import spec
from tests.synth import synth_arrays, class_lists

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def make_args(**kw):
    """Reference defaults (utils/options.py:20-64) for the flags the path reads."""
    a = types.SimpleNamespace(
        batch_size=32, base_lr=3e-5, annotation_num=1, n_classes=5, n_clients=2, local_ep=1,
        device="cpu", rounds_FedMLP_stage1=2, U=0.7, L=0.3, clean_threshold=0.005,
        noise_threshold=0.01, feature_dim=512, model="Resnet18", pretrained=0)
    a.__dict__.update(kw)
    return a


def oracle_net(C, seed):
    from oracle.resnet18_ref import ResNet18Ref
    net = ResNet18Ref(C)
    flat, cnt = spec.init_state("Resnet18", C, seed)
    sd = spec.flat_to_state_dict("Resnet18", C, flat, cnt)
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    return net


def data_dict(n, C, hw, seed, two_view):
    t, x1, x2 = synth_arrays(n, C, hw, seed, two_view)
    d = {"targets": t}
    if two_view:
        d["image_aug_1"], d["image_aug_2"] = torch.from_numpy(x1), torch.from_numpy(x2)
    else:
        d["image"] = torch.from_numpy(x1)
    return d


def norms_of(sd):
    out = {}
    for k, v in sd.items():
        v = torch.as_tensor(np.asarray(v)) if not torch.is_tensor(v) else v
        out[k] = float(torch.linalg.vector_norm(v.double()))
    return out


def assert_norms_close(got, want, rtol, atol=1e-7, what=""):
    for k, w in want.items():
        g = got[k]
        assert abs(g - w) <= atol + rtol * abs(w), f"{what} {k}: got {g} want {w}"


class _MaskedReLU(torch.autograd.Function):
    """relu whose BACKWARD mask is supplied: forward = the oracle's own relu(x)."""

    @staticmethod
    def forward(ctx, x, mask):
        ctx.save_for_backward(mask)
        return x.clamp_min(0)

    @staticmethod
    def backward(ctx, g):
        (mask,) = ctx.saved_tensors
        return g * mask, None


class _RoutedPool(torch.autograd.Function):
    """max-pool whose forward values are the oracle's own and whose backward sends each gradient to a given input position"""

    @staticmethod
    def forward(ctx, x, idx, vals):
        ctx.save_for_backward(idx)
        ctx.shape = x.shape
        return vals.clone()

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        Bn, Cn, H, W = ctx.shape
        gin = torch.zeros((Bn, Cn, H * W), dtype=g.dtype)
        gin.scatter_add_(2, idx.reshape(Bn, Cn, -1), g.reshape(Bn, Cn, -1))
        return gin.view(ctx.shape), None, None


class relu_masks_from_engine:
    """Context manager: the oracle ResNet's backward pass uses the ENGINE's ReLU masks.

    Why: with ~1e6 ReLU inputs per step a few pre-activations always lie within the ~1e-6 distance
    by which two fp32 summation orders differ; such an element's mask is decided by rounding, and one
    flipped mask in a channel with a few hundred values moves that channel's BN-backward sums (and
    everything upstream) by 1e-3.  Taking the masks from the engine's own forward activations
    (fm_debug_activation: mask = value > 0) splits the check into (a) forward parity -- the oracle's
    forward values are untouched and the loss is compared to 1e-5 -- and (b) backward parity for
    identical masks, where 2e-4 holds on every seed.  The number of positions where the engine's
    mask differs from the oracle's own is recorded in `.flips` (it must stay tiny).

    Call order inside the oracle (oracle/resnet18_ref.py): per forward pass the stem ReLU, then for each
    of the 8 basic blocks relu(bn1) and the output ReLU; views are forwarded one after the other.

    stem=True also hands over the stem's two discrete decisions (fm_debug_stem_masks): the ReLU mask of relu(bn1(conv1(x))) and
    the 3x3 max-pool's choice per pooled element -- the oracle's max-pool forward keeps its own values, its backward routes each
    gradient to the position the ENGINE chose (`.pool_flips` counts the differing choices)."""

    def __init__(self, eng, n_views, B, stem=False):
        self.masks = []
        self.pool_codes = []
        stem_m = stem_c = None
        if stem:
            stem_m, stem_c = eng.debug_stem_masks(n_views * B, n_views)
        for v in range(n_views):
            if stem:                                                   # NHWC -> the oracle's NCHW
                self.masks.append(torch.from_numpy(np.ascontiguousarray(stem_m[v * B:(v + 1) * B].transpose(0, 3, 1, 2))))
                self.pool_codes.append(torch.from_numpy(np.ascontiguousarray(stem_c[v * B:(v + 1) * B].transpose(0, 3, 1, 2))))
            else:
                self.masks.append(None)                                # stem: its own mask (6144+ values per channel)
            for blk in range(8):
                for kind in (0, 1):
                    a = eng.debug_activation(kind, blk, n_views * B)[v * B:(v + 1) * B]
                    self.masks.append(torch.from_numpy(a > 0))
        self.flips = 0
        self.calls = 0
        self.pool_flips = 0
        self.pool_calls = 0

    def __enter__(self):
        import torch.nn.functional as F
        self._F, self._orig = F, F.relu
        self._orig_pool = F.max_pool2d

        def max_pool2d(x, *a, **k):
            j = self.pool_calls
            self.pool_calls += 1
            if j >= len(self.pool_codes) or not x.requires_grad:
                return self._orig_pool(x, *a, **k)
            code = self.pool_codes[j].long()                           # kh * 3 + kw of the 3x3 / stride-2 / pad-1 window
            Bn, Cn, H, W = x.shape
            Hp, Wp = code.shape[2], code.shape[3]
            oh = torch.arange(Hp).view(1, 1, Hp, 1)
            ow = torch.arange(Wp).view(1, 1, 1, Wp)
            idx = (2 * oh - 1 + code // 3) * W + (2 * ow - 1 + code % 3)
            own_vals, own_idx = self._orig_pool(x.detach(), 3, 2, 1, return_indices=True)
            self.pool_flips += int((own_idx != idx).sum())
            return _RoutedPool.apply(x, idx, own_vals)
        F.max_pool2d = max_pool2d

        def relu(x, *a, **k):
            j = self.calls
            self.calls += 1
            if j >= len(self.masks) or self.masks[j] is None or not x.requires_grad:
                return self._orig(x, *a, **k)
            m = self.masks[j]
            assert m.shape == x.shape, (j, m.shape, x.shape)
            self.flips += int(((x.detach() > 0) != m).sum())
            return _MaskedReLU.apply(x, m.to(x.dtype))
        F.relu = relu
        return self

    def __exit__(self, *exc):
        self._F.relu = self._orig
        self._F.max_pool2d = self._orig_pool
        return False


def relu_margin(net, xs, max_count=64):
    """Smallest |pre-activation| over the ReLUs whose BN statistics are thin (<= max_count values per
    channel) in a train-mode forward of `net` (a copy: running stats untouched) on the views `xs`.
    A pre-activation within rounding distance of 0 gets its ReLU mask from the summation order of the
    conv that produced it; where a channel holds only a few dozen values, one flipped mask moves that
    channel's BN-backward sums by percents.  Parity tests use this ORACLE-side number to pick
    well-conditioned inputs up front (never by looking at which inputs the HIP path happens to pass)."""
    import copy
    import torch.nn.functional as F
    probe = copy.deepcopy(net).train()
    worst = [float("inf")]
    orig = F.relu

    def spy(t, *a, **k):
        if t.dim() == 4 and t.shape[0] * t.shape[2] * t.shape[3] <= max_count:
            worst[0] = min(worst[0], float(t.detach().abs().min()))
        return orig(t, *a, **k)
    F.relu = spy
    try:
        with torch.no_grad():
            for x in xs:
                probe(x)
    finally:
        F.relu = orig
    return worst[0]


def conditioned_seed(net, make_views, seeds, margin=1e-5):
    """First seed whose inputs keep every thin-statistics ReLU pre-activation of the oracle at least
    `margin` away from 0 (fp32 conv sums differ by ~1e-6 between summation orders)."""
    for sd in seeds:
        if relu_margin(net, make_views(sd)) > margin:
            return sd
    raise AssertionError(f"no seed in {list(seeds)} is {margin}-conditioned")


def replay_local_update():
    """-> ReplayLocalUpdate: fedmlp_amd.local_training.LocalUpdate with the two things a golden replay needs and the
    product class does not carry: recorded batch orders instead of torch.randperm (`order_queue`), and a log of every
    tagging call's pool and similarity row (`tagging_log`).  A factory, so importing tests.helpers needs no GPU library."""
    from fedmlp_amd.local_training import LocalUpdate

    class ReplayLocalUpdate(LocalUpdate):
        def __init__(self, *a, **k):
            super().__init__(*a, **k)
            self.order_queue = []
            self.tagging_log = []

        def _order(self, n):
            if self.order_queue:
                o = list(self.order_queue.pop(0))
                assert len(o) == n
                return o
            return super()._order(n)

        def _log_similarity(self, rnd, cls, pool_idx, sim_of_pool):
            self.tagging_log.append({"rnd": rnd, "cls": cls, "pool_idx": list(pool_idx), "sim": sim_of_pool().cpu().numpy()})

    return ReplayLocalUpdate


def grad_errors(gsd, gold):
    """Engine gradients (state_dict-keyed arrays) against a golden's per-tensor record {norm, sum, head, absmax}
    (tests/golden/make_golden.py::_grad_record).  Per tensor the worst of: |norm - norm_ref| / norm_ref, |sum - sum_ref|
    relative to norm * sqrt(n), and the first three values relative to the tensor's max.  -> (worst tensor, worst, median)
    Tensors whose REFERENCE gradient is numerically zero (norm below 1e-4 of the median tensor's: a BatchNorm bias in front of
    another train-mode BatchNorm has an exactly cancelling gradient, both sides hold rounding noise of ~1e-9 there) are held
    to that floor in absolute terms instead of a ratio of two noises."""
    med_norm = float(np.median([v["norm"] for v in gold.values()]))
    typ = float(np.median([v["absmax"] for v in gold.values()]))
    worst = {}
    for k, w in gold.items():
        got = np.asarray(gsd[k], dtype=np.float64)
        if w["norm"] < 1e-4 * med_norm:
            worst[k] = float(np.linalg.norm(got) / (1e-4 * med_norm)) * 1e-3      # < 1e-3 while the engine's is below the floor too
            continue
        e_norm = abs(np.linalg.norm(got) - w["norm"]) / (w["norm"] + 1e-3 * typ)
        e_sum = abs(got.sum() - w["sum"]) / (w["norm"] * np.sqrt(got.size) + 1e-30)
        e_head = np.abs(got.ravel()[:3] - np.array(w["head"])).max() / (w["absmax"] + 1e-30)
        worst[k] = float(max(e_norm, e_sum, e_head))
    k_bad = max(worst, key=worst.get)
    grad_errors.p90 = float(np.percentile(list(worst.values()), 90))          # (read by the bf16 tests: stated 90th percentile)
    grad_errors.all = dict(worst)
    # the ten worst tensors with their size and their norm relative to the median tensor's (for the parity records)
    grad_errors.top = [[k, round(v, 5), int(np.asarray(gsd[k]).size), round(gold[k]["norm"] / med_norm, 5)]
                       for k, v in sorted(worst.items(), key=lambda kv: -kv[1])[:10]]
    return k_bad, worst[k_bad], float(np.median(list(worst.values())))
