"""The N>1 path on CPU: world_size-2 `gloo` process groups run the same all-reduce
forms the 8-GPU job runs over RCCL (fedmlp_amd/fedavg.py) and are checked against
the single-process reference-surface FedAvg / FedAvg_tao / FedAvg_proto."""
import datetime
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from fedmlp_amd.fedavg import (FedAvg, FedAvg_tao, FedAvg_proto, fedavg_allreduce, tao_allreduce,
                               proto_allreduce)

WORLD = 2
N_LOCAL = [300, 500]
C, D, NSTATE = 4, 16, 1000


class FakeEngine:
    """CPU stand-in with the three engine methods fedavg_allreduce touches."""

    def __init__(self, state, counters):
        self.state, self.cnt = state, counters

    def comm_size(self):
        return 0                      # no library communicator: the sums go through torch.distributed (gloo here)

    def state_scale(self, w):
        self.state.mul_(w)

    def state_tensor(self):
        return self.state

    def counters(self, new=None):
        if new is not None:
            self.cnt = np.asarray(new, dtype=np.int64)
        return self.cnt


def _inputs(rank):
    rs = np.random.RandomState(100 + rank)
    state = torch.from_numpy(rs.standard_normal(NSTATE).astype(np.float32))
    cnt = np.array([40 + 7 * rank, 41 + 7 * rank], dtype=np.int64)
    t = rs.uniform(size=C)
    proto = torch.from_numpy(rs.standard_normal((2 * C, D)).astype(np.float32))
    return state, cnt, t, proto


# client r annotates class r; class 2 has no active client at all, class 3 is missing everywhere
ACT = [[0], [1]]
CLASS_ACTIVE = [[0], [1], [], []]
CLASS_NEG = [[1], [0], [0, 1], [0, 1]]


def _worker(rank, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=WORLD, timeout=datetime.timedelta(seconds=180))
    state, cnt, t, proto = _inputs(rank)
    w = N_LOCAL[rank] / float(sum(N_LOCAL))
    eng = FakeEngine(state.clone(), cnt.copy())
    fedavg_allreduce(eng, w)
    neg_mask = [0.0 if c in ACT[rank] else 1.0 for c in range(C)]
    act_mask = [1.0 if c in ACT[rank] else 0.0 for c in range(C)]
    tao = tao_allreduce(t, N_LOCAL[rank], neg_mask)
    pr = proto_allreduce(proto, N_LOCAL[rank], act_mask)
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), state=eng.state.numpy(), cnt=eng.cnt, tao=tao,
             proto=pr.numpy())
    dist.barrier()
    dist.destroy_process_group()


def _agree_worker(rank, port, out_dir):
    import logging
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=WORLD, timeout=datetime.timedelta(seconds=180))
    from fedmlp_amd.fedavg import state_agreement
    msgs = []

    class H(logging.Handler):
        def emit(self, rec):
            msgs.append(rec.getMessage())
    logging.getLogger("fedmlp_amd.fedavg").addHandler(H())
    state, cnt, _, _ = _inputs(rank)
    eng = FakeEngine(state.clone(), cnt.copy())
    before = state_agreement(eng)                       # different inputs per rank: must be seen
    fedavg_allreduce(eng, 0.5)
    fedavg_allreduce(eng, 0.5)
    after = state_agreement(eng)
    eng.state[rank * 7 + 3] += 1e-6 * (rank + 1)        # ONE element nudged differently on each rank
    nudged = state_agreement(eng)
    np.savez(os.path.join(out_dir, f"a{rank}.npz"), before=before[0], after=after[0], nudged=nudged[0], worst=nudged[1],
             warnings=len([m for m in msgs if "torch.distributed" in m]))
    dist.barrier()
    dist.destroy_process_group()


def test_state_agreement_and_fallback_warning(tmp_path):
    """after an all-reduce the ranks' states are compared through one extra collective of checksums
    (bench.py --gpus N and driver.py fail on disagreement), and the torch.distributed fallback of fedavg_allreduce says so ONCE."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_agree_worker, args=(port, str(tmp_path)), nprocs=WORLD, join=True)
    for r in range(WORLD):
        got = np.load(os.path.join(str(tmp_path), f"a{r}.npz"))
        assert not bool(got["before"]) and bool(got["after"]) and not bool(got["nudged"])
        assert float(got["worst"]) > 0
        assert int(got["warnings"]) == 1                # two all-reduces, one warning


def test_allreduce_forms_match_reference_surface(tmp_path):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_worker, args=(port, str(tmp_path)), nprocs=WORLD, join=True)
    ins = [_inputs(r) for r in range(WORLD)]
    # single-process reference surface on the same inputs
    sds = [{"w": ins[r][0], "bn.num_batches_tracked": torch.tensor(ins[r][1][0])} for r in range(WORLD)]
    want = FedAvg(sds, N_LOCAL)
    want_tao = FedAvg_tao([ins[r][2] for r in range(WORLD)], N_LOCAL, CLASS_NEG)
    want_proto = FedAvg_proto([ins[r][3] for r in range(WORLD)], N_LOCAL, CLASS_ACTIVE).numpy()
    for r in range(WORLD):
        got = np.load(os.path.join(str(tmp_path), f"r{r}.npz"))
        np.testing.assert_allclose(got["state"], want["w"].numpy(), rtol=2e-6, atol=1e-7)
        assert got["cnt"][0] == int(np.trunc(float(want["bn.num_batches_tracked"])))
        np.testing.assert_allclose(got["tao"], want_tao, rtol=1e-12)
        np.testing.assert_allclose(got["proto"], want_proto, rtol=2e-6, atol=1e-7, equal_nan=True)
        assert np.isnan(got["proto"][4:]).all()          # classes without an active client


# ---- comm_init: every rank must end up on the same path -------------------------------------------------
class FakeCommEngine:
    """stand-in with the four communicator methods fedavg.comm_init touches"""
    device = "cpu"

    def __init__(self, fail_init=False, fail_id=False, fail_preflight=False):
        self.fail_init, self.fail_id, self.size, self.destroyed = fail_init, fail_id, 0, False
        self.fail_preflight, self.entered_init = fail_preflight, False

    def comm_preflight(self):
        if self.fail_preflight:
            raise RuntimeError("cannot load librccl")

    def comm_unique_id(self):
        if self.fail_id:
            raise RuntimeError("no RCCL")
        return bytes(range(128))

    def comm_init(self, uid, rank, world):
        assert uid == bytes(range(128))
        self.entered_init = True
        if self.fail_init:
            raise RuntimeError("ncclCommInitRank failed")
        self.size = world

    def comm_destroy(self):
        self.destroyed, self.size = True, 0

    def comm_size(self):
        return self.size


def _comm_worker(rank, port, out_dir, case):
    from fedmlp_amd.fedavg import comm_init
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=WORLD, timeout=datetime.timedelta(seconds=180))
    eng = FakeCommEngine(fail_init=(case == "init_fails_on_rank1" and rank == 1), fail_id=(case == "id_fails" and rank == 0),
                         fail_preflight=(case == "preflight_fails_on_rank1" and rank == 1))
    try:
        res = ("ok", comm_init(eng))
    except RuntimeError as ex:
        res = ("raised", str(ex))
    with open(os.path.join(out_dir, f"c{rank}.txt"), "w") as f:
        f.write(f"{res[0]}|{res[1]}|{eng.comm_size()}|{int(eng.destroyed)}|{int(eng.entered_init)}")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("case", ["all_fine", "init_fails_on_rank1", "id_fails", "preflight_fails_on_rank1"])
def test_comm_init_failure_is_agreed_on_by_all_ranks(tmp_path, case):
    """A library communicator that forms on some ranks only would leave the ranks on different all-reduce paths
    (deadlock): comm_init either returns the rank count everywhere or raises everywhere, and a rank whose own
    fm_comm_init succeeded drops its communicator again."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_comm_worker, args=(port, str(tmp_path), case), nprocs=WORLD, join=True)
    res = [open(os.path.join(str(tmp_path), f"c{r}.txt")).read().split("|") for r in range(WORLD)]
    if case == "all_fine":
        assert [r[0] for r in res] == ["ok", "ok"] and [r[2] for r in res] == ["2", "2"]
    else:
        assert [r[0] for r in res] == ["raised", "raised"], res
        assert [r[2] for r in res] == ["0", "0"], res               # nobody keeps a communicator
        if case == "init_fails_on_rank1":
            assert res[0][3] == "1"                                  # rank 0 had one and destroyed it
        if case == "preflight_fails_on_rank1":
            # librccl missing on one rank: NO rank may enter the collective ncclCommInitRank (its peers would block in it)
            assert [r[4] for r in res] == ["0", "0"], res
