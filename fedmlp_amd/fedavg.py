"""Aggregation: the reference's utils/FedAvg.py surface + its RCCL form.

Two forms of the same arithmetic:
  * FedAvg / FedAvg_tao / FedAvg_proto -- drop-ins with the reference signatures
    (utils/FedAvg.py:7-14, 51-70, 72-93) for a single-process driver that holds
    every client's state_dict on the host, as main.py:216-234 does.  Host glue:
    left-to-right weighted mean, same order as the reference.
  * fedavg_allreduce / tao_allreduce / proto_allreduce -- one client per GPU:
    each rank pre-scales its device-resident state by n_i/sum(n) (HIP kernel) and
    the sum is an RCCL all-reduce over xGMI (torch.distributed backend "nccl");
    on CPU test rigs the same code runs over gloo tensors.
"""
import copy
import logging
from collections import OrderedDict

import numpy as np
import torch


def _np(v):
    return v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)


def FedAvg(w, dict_len):
    """utils/FedAvg.py:7-14: sample-count weighted mean of EVERY state_dict entry,
    accumulated left to right in the entry's dtype; the int64 counter becomes
    float through the true division (and is truncated again on load)."""
    out = OrderedDict()
    for k in w[0].keys():
        acc = _np(w[0][k]) * dict_len[0]
        for i in range(1, len(w)):
            acc = acc + _np(w[i][k]) * dict_len[i]
        if np.issubdtype(acc.dtype, np.integer):
            acc = (acc / float(sum(dict_len))).astype(np.float32)
        else:
            acc = (acc / np.float32(sum(dict_len))).astype(np.float32)
        out[k] = torch.from_numpy(np.ascontiguousarray(acc))
    return out


def FedAvg_tao(t, weight, class_active_client_list=None):
    """utils/FedAvg.py:51-70."""
    C = len(t[0])
    out = np.array([0.] * C)
    if class_active_client_list is None:
        for i, tao in enumerate(t):
            out += np.asarray(tao) * float(weight[i])
        return out / float(sum(weight))
    for cls, clients in enumerate(class_active_client_list):
        if len(clients) == 0:
            out[cls] = 1.
            continue
        wsum = 0.
        for i, tao in enumerate(t):
            if i in clients:
                out[cls] += tao[cls] * float(weight[i])
                wsum += float(weight[i])
        out[cls] = out[cls] / wsum
    return out


def FedAvg_proto(Prototypes, weight, class_active_client_list):
    """utils/FedAvg.py:72-93 (a class with no active client yields NaN rows)."""
    P = [torch.as_tensor(_np(p)) for p in Prototypes]
    out = torch.zeros((len(P[0]), len(P[0][0])))
    for cls, clients in enumerate(class_active_client_list):
        a0 = torch.zeros_like(P[0][0])
        a1 = torch.zeros_like(P[0][0])
        for cid in clients:
            a0 = P[cid][2 * cls] * weight[cid] + a0
            a1 = P[cid][2 * cls + 1] * weight[cid] + a1
        den = np.sum(np.array(weight)[clients])
        out[2 * cls] = a0 / den
        out[2 * cls + 1] = a1 / den
    return out


# ---- one client per GPU: RCCL --------------------------------------------------------
_log = logging.getLogger("fedmlp_amd.fedavg")
_warned_fallback = False


def _dist():
    import torch.distributed as dist
    return dist if (dist.is_available() and dist.is_initialized()) else None


def allreduce_weighted_(tensor, w):
    """tensor <- sum_ranks(w_rank * tensor_rank), in place (tensor on any device)."""
    tensor.mul_(w)
    d = _dist()
    if d is not None and d.get_world_size() > 1:
        d.all_reduce(tensor, op=d.ReduceOp.SUM)
    return tensor


def comm_init(engine):
    """Create the library's own RCCL communicator (fm_comm_init) over the ranks of the current
    torch.distributed job: rank 0 draws the ncclUniqueId, torch.distributed only carries those
    128 bytes.  Returns the communicator's rank count."""
    d = _dist()
    if d is None or d.get_world_size() == 1:
        return 1
    # Every rank must end up on the SAME path (library communicator or torch.distributed fallback), so failures are
    # agreed on: rank 0 ships None when it cannot draw an id, and after fm_comm_init the ranks take the minimum of
    # their success flags; if any rank failed, those that succeeded drop their communicator and all of them raise.
    # (0) non-collective preflight, agreed on before any rank enters ncclCommInitRank: a rank that cannot load librccl
    # would otherwise leave its peers blocked inside the collective init, never reaching the agreement below
    pre = None
    try:
        engine.comm_preflight()
    except Exception as ex:                                  # noqa: BLE001
        pre = ex
    ok0 = torch.tensor([0 if pre is not None else 1], dtype=torch.int32, device=engine.device)
    d.all_reduce(ok0, op=d.ReduceOp.MIN)
    if int(ok0.item()) == 0:
        raise RuntimeError(f"fm_comm_preflight failed on at least one rank ({pre})")
    box = [None]
    if d.get_rank() == 0:
        try:
            box = [engine.comm_unique_id()]
        except Exception:                                    # noqa: BLE001
            box = [None]
    d.broadcast_object_list(box, src=0)
    if box[0] is None:
        raise RuntimeError("fm_comm_unique_id failed on rank 0")
    err = None
    try:
        engine.comm_init(box[0], d.get_rank(), d.get_world_size())
    except Exception as ex:                                  # noqa: BLE001
        err = ex
    ok = torch.tensor([0 if err is not None else 1], dtype=torch.int32, device=engine.device)
    d.all_reduce(ok, op=d.ReduceOp.MIN)
    if int(ok.item()) == 0:
        if err is None:
            engine.comm_destroy()
        raise RuntimeError(f"fm_comm_init failed on at least one rank ({err})")
    return engine.comm_size()


def fedavg_allreduce(engine, w):
    """FedAvg of the device-resident model over all ranks: state <- sum_i w_i state_i with
    w_i = n_i / sum(n).  With a library communicator (comm_init) the pre-scale, the
    ncclAllReduce of the arena and the counter mean all run inside the C ABI on the engine's
    stream (fm_fedavg_allreduce); otherwise the sum goes through torch.distributed."""
    if engine.comm_size() >= 1:
        engine.fedavg_allreduce(float(w))
        return engine.state_tensor()
    d = _dist()
    engine.state_scale(float(w))
    st = engine.state_tensor()
    if d is not None and d.get_world_size() > 1:
        global _warned_fallback
        if not _warned_fallback:
            # a caller that skipped comm_init (or whose comm_init failed) gets a DIFFERENT path from the one bench.py / driver.py
            # insist on: say so once
            _warned_fallback = True
            _log.warning("fedavg_allreduce: no library RCCL communicator (fm_comm_init was not called or failed): the all-reduce "
                         "of the state goes through torch.distributed (%s) instead of fm_fedavg_allreduce", d.get_backend())
        d.all_reduce(st, op=d.ReduceOp.SUM)
        engine.state_tensor()                 # marks the engine's derived buffers (BN folds, weight packs) stale
        cnt = torch.from_numpy(engine.counters().astype(np.float64) * float(w)).to(st.device)
        d.all_reduce(cnt, op=d.ReduceOp.SUM)
        engine.counters(np.trunc(cnt.cpu().numpy() + 1e-9).astype(np.int64))
    return st


def state_agreement(engine):
    """After a FedAvg every rank must hold the SAME state (utils/FedAvg.py:7-14 hands one w_glob to every client, main.py:216-222).
    Three checksums of the device arena -- the sum of its bit patterns as integers (any flipped bit moves it), the float64 sum and
    the float64 sum of magnitudes -- are compared across ranks with ONE extra all-reduce: returns (agree, worst) where worst is the
    largest difference between the ranks' checksums, relative to the magnitude sum.  A world of one agrees by definition.
    The first multi-GPU run of bench.py / driver.py is therefore also a correctness run of the RCCL path."""
    d = _dist()
    if d is None or d.get_world_size() <= 1:
        return True, 0.0
    st = engine.state_tensor()
    bits = st.view(torch.int32).to(torch.int64).sum().to(torch.float64)
    s1 = st.to(torch.float64).sum()
    s2 = st.to(torch.float64).abs().sum()
    v = torch.stack([bits, s1, s2, -bits, -s1, -s2])
    d.all_reduce(v, op=d.ReduceOp.MAX)                 # max(x) and max(-x) = -min(x) in one collective
    v = v.cpu().numpy()
    spread = v[:3] + v[3:]                             # max - min of each checksum over the ranks
    scale = max(abs(float(v[2])), 1e-30)
    worst = float(max(spread[1], spread[2]) / scale)
    return bool(spread[0] == 0.0 and spread[1] == 0.0 and spread[2] == 0.0), worst


def tao_allreduce(t, n_i, is_negative_client_mask, device="cpu", engine=None):
    """FedAvg_tao over ranks: t[C] local, mask[c] = 1 if this client has class c missing
    (it is in class_negative_client_list[c], main.py:206-210, 223)."""
    if engine is not None and engine.comm_size() >= 1:
        return engine.fedavg_tao(t, n_i, is_negative_client_mask)
    m = np.asarray(is_negative_client_mask, dtype=np.float64)
    buf = torch.from_numpy(np.concatenate([np.asarray(t, np.float64) * n_i * m, n_i * m])).to(device)
    d = _dist()
    if d is not None and d.get_world_size() > 1:
        d.all_reduce(buf, op=d.ReduceOp.SUM)
    buf = buf.cpu().numpy()
    C = len(m)
    num, den = buf[:C], buf[C:]
    return np.where(den == 0, 1.0, num / np.where(den == 0, 1.0, den))


def proto_allreduce(proto, n_i, is_active_client_mask, device="cpu", engine=None):
    """FedAvg_proto over ranks: proto[2C,D] local, mask[c] = 1 if this client annotates c."""
    if engine is not None and engine.comm_size() >= 1:
        return torch.from_numpy(engine.fedavg_proto(_np(proto), n_i, is_active_client_mask))
    m = np.repeat(np.asarray(is_active_client_mask, dtype=np.float32), 2)[:, None]
    P = np.asarray(_np(proto), dtype=np.float32) * (np.float32(n_i) * m)     # NaN rows of an active class propagate
    num = torch.from_numpy(np.where(m > 0, P, 0.0).astype(np.float32)).to(device)
    den = torch.from_numpy((np.float32(n_i) * m[:, 0]).astype(np.float32)).to(device)
    d = _dist()
    if d is not None and d.get_world_size() > 1:
        d.all_reduce(num, op=d.ReduceOp.SUM)
        d.all_reduce(den, op=d.ReduceOp.SUM)
    with np.errstate(invalid="ignore", divide="ignore"):
        return torch.from_numpy(num.cpu().numpy() / den.cpu().numpy()[:, None])   # 0/0 -> NaN


def consistency_weight(rnd, begin, end):
    """get_current_consistency_weight(rnd, args.begin, args.end) (utils/FedNoRo.py:72-81): the
    sigmoid ramp-up main.py:127-128 multiplies by args.a to get train_FedNoRo's weight_kd."""
    cur = float(np.clip(rnd, begin, end))
    phase = 1.0 - (cur - begin) / (end - begin)
    return float(np.exp(-5.0 * phase * phase))
