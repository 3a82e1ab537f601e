"""Known-answer tests of the index / selection / prototype work ON THE GPU, through the C ABI,
straight from the reference-generated vectors of tests/golden/kat.json (made by
tests/golden/make_golden.py importing /root/reference): stable top-/bottom-k with ties
(utils/utils.py:24-35 via utils/local_training.py:1061-1087) bit-exact, CosineSimilarityFast
(:1417-1435) to 1e-6, the prototype + t pass (:971-1002, :1208-1250) at D = 512 including the
0/0 = NaN and zero-guard cases.  Plus the library's own RCCL entry points on a world of one."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import steps_ref as R
from tests.helpers import load_golden

pytestmark = pytest.mark.gpu

C_, HW, D = 5, 64, 512


@pytest.fixture(scope="module")
def eng():
    from fedmlp_amd.engine import Engine
    e = Engine("Resnet18", C_, HW, HW, 16)
    yield e
    e.close()


@pytest.fixture(scope="module")
def kat():
    return load_golden("kat.json")


def _select(eng, values, clean_thr, noise_thr):
    sim = torch.tensor(values, dtype=torch.float32, device=eng.device)
    return eng.select_topk(sim, clean_thr, noise_thr)


def test_select_topk_reference_kat_with_ties(eng, kat):
    """kat.topk.lst = [.1, .5, .5, -1, .5, -1, 3, .1]: three-way and two-way ties.  n_clean = 6,
    n_noise = 2; thresholds are chosen so that int(thr * n) walks every k the golden holds."""
    g = kat["topk"]
    lst = g["lst"]
    n_clean = sum(v >= 0 for v in lst)
    n_noise = sum(v < 0 for v in lst)
    for kt in range(n_clean + 1):
        for kb in range(n_noise + 1):
            top, bot = _select(eng, lst, (kt + 0.5) / n_clean, (kb + 0.5) / n_noise)
            assert top == g["max"][str(kt)], (kt, top)
            assert bot == g["min"][str(kb)], (kb, bot)


def test_select_topk_heavy_ties_matches_reference_semantics(eng):
    """N = 5 000 similarities quantised to 41 levels (every value tied ~120 times) and an
    all-equal row: positions must equal Python's stable sorted() order exactly."""
    rs = np.random.RandomState(5)
    for vals in (np.round(rs.uniform(-1, 1, 5000) * 20) / 20, np.zeros(777), -np.ones(300)):
        vals = vals.astype(np.float32)
        lst = vals.tolist()
        n_clean = int((vals >= 0).sum()); n_noise = int((vals < 0).sum())
        for ct, nt in ((0.005, 0.01), (0.25, 0.5), (1.0, 1.0)):
            kt, kb = int(1 * ct * n_clean), int(1 * nt * n_noise)
            top, bot = _select(eng, lst, ct, nt)
            assert top == R.max_m_indices(lst, kt)
            assert bot == R.min_n_indices(lst, kb)


def test_select_topk_empty_and_single(eng):
    assert _select(eng, [0.25], 0.005, 0.01) == ([], [])
    assert _select(eng, [0.25], 1.0, 1.0) == ([0], [])
    assert _select(eng, [-0.25], 1.0, 1.0) == ([], [0])
    sim = torch.empty(0, dtype=torch.float32, device=eng.device)
    assert eng.select_topk(sim, 1.0, 1.0) == ([], [])


def _pad(a, width):
    out = np.zeros((a.shape[0], width), np.float32)
    out[:, :a.shape[1]] = a
    return out


def test_cos_tag_reference_kat(eng, kat):
    """kat.cosine (D = 16, embedded in the engine's D = 512 with zero columns, which leaves every
    dot product and norm unchanged): sim = cos(f, P0) - cos(f, P1) to 1e-6."""
    g = kat["cosine"]
    f = _pad(np.asarray(g["f"], np.float32), D)
    proto = np.zeros((2 * C_, D), np.float32)
    cls = 3
    proto[2 * cls, :16] = np.asarray(g["p0"], np.float32)
    proto[2 * cls + 1, :16] = np.asarray(g["p1"], np.float32)
    sim = eng.cos_tag(torch.from_numpy(f).to(eng.device), torch.from_numpy(proto).to(eng.device), [cls])
    np.testing.assert_allclose(sim[0].cpu().numpy(), np.asarray(g["sim"], np.float32), rtol=0, atol=1e-6)
    # SURVEY's hand-checkable triple
    f3 = _pad(np.array([[1., 2, 3], [0, -1, .5], [2, 0, 0]], np.float32), D)
    p3 = np.zeros((2 * C_, D), np.float32)
    p3[0, :3] = [1., 0, 0]; p3[1, :3] = [0., 1, 1]
    sim = eng.cos_tag(torch.from_numpy(f3).to(eng.device), torch.from_numpy(p3).to(eng.device), [0])
    np.testing.assert_allclose(sim[0].cpu().numpy(), np.asarray(kat["cosine_survey"]["sim"], np.float32), atol=1e-6)


def test_cos_tag_nan_prototype_disables_class(eng):
    """FedAvg_proto leaves NaN rows for a class without an active client (utils/FedAvg.py:85-86):
    sim is NaN, counts as neither >= 0 nor < 0, so nothing is selected."""
    rs = np.random.RandomState(1)
    f = torch.from_numpy(rs.standard_normal((300, D)).astype(np.float32)).to(eng.device)
    proto = torch.from_numpy(rs.standard_normal((2 * C_, D)).astype(np.float32))
    proto[4:6] = float("nan")
    sim = eng.cos_tag(f, proto.to(eng.device), [2, 1])
    assert torch.isnan(sim[0]).all() and torch.isfinite(sim[1]).all()
    assert eng.select_topk(sim[0].contiguous(), 1.0, 1.0) == ([], [])
    want = R.cosine_diff(f.cpu(), proto[2], proto[3])
    np.testing.assert_allclose(sim[1].cpu().numpy(), np.asarray(want), atol=1e-6)


def test_proto_pass_d512_matches_oracle_incl_nan_and_zero_guard(eng):
    """Two accumulate calls over ragged batches at D = 512; class 0 has positives and negatives,
    class 2 is active with NO positives: unguarded finalisation gives a 0/0 = NaN row
    (utils/local_training.py:997-999), the guarded one leaves the zero row (:1240-1248)."""
    rs = np.random.RandomState(9)
    batches = []
    for b in (7, 3):
        f = torch.from_numpy(rs.standard_normal((b, D)).astype(np.float32))
        z = torch.from_numpy(rs.standard_normal((b, C_)).astype(np.float32) * 2)
        y = torch.from_numpy((rs.uniform(size=(b, C_)) < 0.4).astype(np.float32))
        y[:, 2] = 0.0
        batches.append((f, z, y))
    act_list, neg_list = [0, 2], [1, 3, 4]
    act = [1.0 if c in act_list else 0.0 for c in range(C_)]
    neg = [1.0 if c in neg_list else 0.0 for c in range(C_)]
    n_local = 10
    for guard in (False, True):
        eng.proto_reset()
        for f, z, y in batches:
            eng.proto_accumulate(f.to(eng.device), z.to(eng.device), y.to(eng.device), act, neg, 0.3, 0.7)
        t, proto = eng.proto_finalize(guard, n_local, act)
        want_t, want_p = R.prototype_pass(batches, C_, act_list, neg_list, 0.3, 0.7, n_local, guard)
        np.testing.assert_array_equal(t, want_t)                       # integer counts / n: exact
        np.testing.assert_allclose(proto, want_p.numpy(), rtol=1e-6, atol=1e-6, equal_nan=True)
        assert np.isnan(proto[5]).all() != guard                       # row 2*2+1: NaN unguarded, zeros guarded
        assert (proto[2] == 0).all() and (proto[6:] == 0).all()        # non-active rows stay zero


def test_library_rccl_world_of_one(eng):
    """fm_comm_unique_id / fm_comm_init / fm_fedavg_* through the C ABI with one rank: RCCL loads,
    the communicator forms, and the all-reduce of the state arena is the identity on w = 1."""
    from fedmlp_amd import spec
    flat, cnt = spec.init_state("Resnet18", C_, 11)
    cnt = cnt + 3
    eng.set_state(flat, cnt)
    uid = eng.comm_unique_id()
    assert len(uid) == 128 and any(uid)
    eng.comm_init(uid, 0, 1)
    assert eng.comm_size() == 1
    eng.fedavg_allreduce(1.0)
    got, gcnt = eng.get_state()
    np.testing.assert_array_equal(got, flat)
    np.testing.assert_array_equal(gcnt, cnt)
    eng.fedavg_allreduce(0.5)
    got, _ = eng.get_state()
    np.testing.assert_array_equal(got, flat * np.float32(0.5))
    t = eng.fedavg_tao([0.1, 0.2, 0.3, 0.4, 0.5], 300, [0, 1, 1, 0, 1])
    np.testing.assert_allclose(t, [1.0, 0.2, 0.3, 1.0, 0.5], rtol=1e-15)
    p = np.arange(2 * C_ * D, dtype=np.float32).reshape(2 * C_, D)
    out = eng.fedavg_proto(p, 300, [1, 0, 0, 0, 0])
    np.testing.assert_allclose(out[:2], p[:2], rtol=1e-6)
    assert np.isnan(out[2:]).all()


def test_fedavg_fold_is_bit_exact_with_the_reference(eng, kat):
    """fm_fedavg_fold (several clients' states on ONE GPU) reproduces utils/FedAvg.py:7-14 bit for bit on fp32 entries: the
    reference's own known-answer vector embedded at the head of arena-sized states, and the rest of the arena against the
    same left-to-right float32 arithmetic in numpy (what the host drop-in FedAvg, pinned on that KAT, computes)."""
    g = kat["fedavg"]
    n = eng.state_tensor().numel()
    lens = g["lens"]
    K = len(g["w"])
    rs = np.random.RandomState(5)
    host = [rs.standard_normal(n).astype(np.float32) for _ in range(K)]
    fkeys = [k for k in g["w"][0] if "num_batches" not in k]
    want_head = np.concatenate([np.asarray(g["out"][k], np.float32).reshape(-1) for k in fkeys])
    for i in range(K):
        head = np.concatenate([np.asarray(g["w"][i][k], np.float32).reshape(-1) for k in fkeys])
        host[i][:head.size] = head
    states = [torch.from_numpy(h).to(eng.device) for h in host]
    out = torch.empty_like(states[0])
    eng.fedavg_fold(states, lens, out)
    got = out.cpu().numpy()
    np.testing.assert_array_equal(got[:want_head.size], want_head)
    acc = host[0] * np.float32(lens[0])
    for i in range(1, K):
        acc = acc + host[i] * np.float32(lens[i])
    np.testing.assert_array_equal(got, acc / np.float32(sum(lens)))


def test_fedavg_fold_of_eight_clients_into_the_engine_state(eng):
    """K = 8 (BASELINE configs[1]: 8 clients), uneven sample counts, the engine's own state as the destination: the folded
    state is what the next forward uses (derived buffers are rebuilt)."""
    from fedmlp_amd import spec
    flat, cnt = spec.init_state("Resnet18", C_, 1037)
    eng.set_state(flat, cnt)
    base = eng.state_tensor().clone()
    g = torch.Generator(device=eng.device).manual_seed(2)
    lens = [5000, 4999, 37, 5000, 1, 2500, 5000, 123]
    states = [(base * (1.0 + 0.01 * torch.randn(base.shape, device=eng.device, generator=g))).contiguous() for _ in lens]
    hs = [s.cpu().numpy() for s in states]
    acc = hs[0] * np.float32(lens[0])
    for i in range(1, len(lens)):
        acc = acc + hs[i] * np.float32(lens[i])
    want = acc / np.float32(sum(lens))
    x = torch.randn((4, 3, HW, HW), device=eng.device, generator=g)
    eng.forward_eval(x)                                   # derived buffers (BN folds, packs) built for the old state
    eng.fedavg_fold(states, lens)                         # out = the engine's state
    np.testing.assert_array_equal(eng.state_tensor().cpu().numpy(), want)
    f1, z1 = eng.forward_eval(x)
    folded_flat, _ = eng.get_state()
    eng.set_state(folded_flat, cnt)
    f2, z2 = eng.forward_eval(x)
    assert torch.equal(z1, z2) and torch.equal(f1, f2)
    with pytest.raises(Exception):
        eng.fedavg_fold(states * 3, lens * 3)             # K > 16


def test_select_topk_rows_matches_per_class_selection(eng):
    """fm_select_topk_rows (every class of a round in one launch pair, one device-to-host read) returns, class by class, exactly
    what fm_select_topk returns on that class's pool: whole rows and sub-pools in arbitrary pool order, heavy ties, a NaN row
    (a class nobody annotates: no pick), an empty pool."""
    rng = np.random.RandomState(5)
    N, n_cls = 777, 6
    sims = np.round(rng.randn(n_cls, N).astype(np.float32), 1)          # one decimal: many exact ties
    sims[3] = np.nan
    sims_d = torch.from_numpy(sims).to(eng.device)
    pools = [None, list(rng.permutation(N)[:300]), list(rng.permutation(N)[:41]), None, [], list(range(N - 1, -1, -1))]
    for ct, nt in ((0.005, 0.01), (0.3, 0.5), (1.0, 1.0)):
        got = eng.select_topk_rows(sims_d, pools, ct, nt)
        assert len(got) == n_cls
        for k in range(n_cls):
            rows = list(range(N)) if pools[k] is None else [int(r) for r in pools[k]]
            want = eng.select_topk(torch.from_numpy(sims[k][rows]).to(eng.device).contiguous(), ct, nt) if rows else ([], [])
            assert got[k] == want, (k, ct, nt)
    assert eng.select_topk_rows(sims_d[:0], [], 0.5, 0.5) == []
