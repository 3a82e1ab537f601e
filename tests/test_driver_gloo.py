"""driver.py's per-round aggregation (RoundAccumulator: main.py:216-234 split over ranks) end to end on CPU:
4 clients on 2 ranks (gloo, world size 2; each rank folds its two clients first), against the 1-rank run of the same
code and against the reference-surface FedAvg / FedAvg_tao / FedAvg_proto (pinned on the reference's KATs in
test_oracle_golden.py) -- state, num_batches_tracked, tao and prototypes incl. the NaN rows of a class nobody annotates,
the 1.0 tao of a class nobody misses and a NaN prototype row of an ACTIVE class (0/0 of the unguarded first pass)."""
import datetime
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from fedmlp_amd.driver import RoundAccumulator
from fedmlp_amd.fedavg import FedAvg, FedAvg_tao, FedAvg_proto
from tests.test_fedavg_gloo import FakeEngine

N_CLIENTS, C, D, NSTATE = 4, 6, 16, 1000
N_LOCAL = [300, 500, 200, 400]
# client c annotates ACT[c]; class 4 has no active client (NaN prototype rows), class 5 is annotated by everyone
# (nobody misses it: tao = 1.0), class 0 by two clients
ACT = [[0, 5], [1, 5], [2, 0, 5], [3, 5]]


def _client(c):
    rs = np.random.RandomState(500 + c)
    state = torch.from_numpy(rs.standard_normal(NSTATE).astype(np.float32))
    cnt = np.array([40 + 7 * c, 41 + 3 * c], dtype=np.int64)
    t = rs.uniform(size=C)
    proto = torch.from_numpy(rs.standard_normal((2 * C, D)).astype(np.float32))
    if c == 2:
        proto[2 * 2 + 1] = float("nan")          # no positive sample of its own class: 0/0 in the first prototype pass
    return state, cnt, t, proto


def _run_rank(rank, world, with_tp=True):
    glob = torch.zeros(NSTATE)
    acc = RoundAccumulator(glob, 2, C, D, float(sum(N_LOCAL)))
    for c in range(N_CLIENTS):
        if c % world != rank:
            continue
        state, cnt, t, proto = _client(c)
        ret = (None, 0.0, None, None, None, None, t, proto)
        acc.add(N_LOCAL[c], state, cnt, ACT[c], ret)
    eng = FakeEngine(torch.zeros(NSTATE), np.zeros(2, np.int64))
    cnt, tao, proto = acc.reduce(eng, "cpu", with_tao_proto=with_tp)
    return eng.state.numpy().copy(), cnt, tao, proto.numpy()


def _worker(rank, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=2, timeout=datetime.timedelta(seconds=180))
    state, cnt, tao, proto = _run_rank(rank, 2)
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), state=state, cnt=cnt, tao=tao, proto=proto)
    dist.barrier()
    dist.destroy_process_group()


def test_four_clients_on_two_ranks_equal_one_rank_and_the_reference_surface(tmp_path):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_worker, args=(port, str(tmp_path)), nprocs=2, join=True)
    one = _run_rank(0, 1)                                       # the same code, every client on one rank, no process group
    cl = [_client(c) for c in range(N_CLIENTS)]
    sds = [{"w": cl[c][0], "bn.num_batches_tracked": torch.tensor(cl[c][1][0])} for c in range(N_CLIENTS)]
    want = FedAvg(sds, N_LOCAL)
    cls_act = [[c for c in range(N_CLIENTS) if k in ACT[c]] for k in range(C)]
    cls_neg = [[c for c in range(N_CLIENTS) if k not in ACT[c]] for k in range(C)]
    want_tao = FedAvg_tao([cl[c][2] for c in range(N_CLIENTS)], N_LOCAL, cls_neg)
    want_proto = FedAvg_proto([cl[c][3] for c in range(N_CLIENTS)], N_LOCAL, cls_act).numpy()
    assert want_tao[5] == 1.0 and np.isnan(want_proto[8:10]).all() and np.isnan(want_proto[5]).all()
    for r in range(2):
        got = np.load(os.path.join(str(tmp_path), f"r{r}.npz"))
        for name, g, o, w in (("state", got["state"], one[0], want["w"].numpy()),
                              ("tao", got["tao"], one[2], want_tao),
                              ("proto", got["proto"], one[3], want_proto)):
            assert np.array_equal(np.isnan(g), np.isnan(w)), name
            np.testing.assert_allclose(g, o, rtol=1e-6, atol=1e-6, equal_nan=True, err_msg=name + " vs 1 rank")
            np.testing.assert_allclose(g, w, rtol=1e-6, atol=1e-6, equal_nan=True, err_msg=name + " vs reference surface")
        assert got["cnt"][0] == one[1][0] == int(np.trunc(float(want["bn.num_batches_tracked"])))
