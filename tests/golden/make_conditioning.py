#!/usr/bin/env python3
"""Measure how well-conditioned the golden trajectories are, with the CPU oracle only.

Adam's first updates are +-lr*sign(g): for zero-initialised BN biases whose gradient is
~0, a 1e-7 relative perturbation of the inputs flips update directions, so after a round
their (tiny) norms differ by ~1e-2 relative while every weight tensor agrees to ~1e-5.
This script replays client 0's first stage-1 round of tests/golden/traj_fedmlp.json three
ways -- 8 threads, 1 thread (different fp32 summation order), and with all weights scaled
by (1 + 1e-7) -- and records the per-tensor norm deviations.  The GPU parity tests use
3x these measured deviations (floor 1e-3) as their per-tensor tolerance.

usage: python tests/golden/make_conditioning.py   -> tests/golden/conditioning.json
"""
import copy
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import steps_ref as R                                      # noqa: E402
from tests.helpers import load_golden, make_args, oracle_net, data_dict, norms_of  # noqa: E402
from tests.synth import class_lists                                    # noqa: E402

g = load_golden("traj_fedmlp.json")
C, n_cl, N, S1 = g["C"], g["n_clients"], g["N"], g["S1"]
args = make_args(n_classes=C, n_clients=n_cl, rounds_FedMLP_stage1=S1)
data = data_dict(n_cl * N, C, g["hw"], g["data_seed"], True)
_, neg = class_lists(data["targets"], C)
res = {}
for name, nt, pert in (("base", 8, 0.0), ("threads1", 1, 0.0), ("perturb1e-7", 8, 1e-7)):
    torch.set_num_threads(nt)
    glob = oracle_net(C, g["init_seed"])
    if pert:
        with torch.no_grad():
            for p in glob.parameters():
                p.mul_(1.0 + pert)
    cl = R.RefClient(args, 0, data, g["users"][0], neg, [0])
    ret = cl.stage1(copy.deepcopy(glob), g["rounds"][0]["train_orders"][0], with_proto=False)
    res[name] = (ret[1], norms_of(ret[0]))
base = res["base"]
out = {"what": "relative deviation of per-tensor L2 norms / mean loss after ONE stage-1 round "
               "(16 Adam steps) of the CPU oracle under benign perturbations",
       "loss": {}, "norms": {}}
for name in ("threads1", "perturb1e-7"):
    l, n = res[name]
    out["loss"][name] = abs(l - base[0]) / base[0]
    for k in n:
        if "num_batches" in k:
            continue
        d = abs(n[k] - base[1][k]) / (abs(base[1][k]) + 1e-12)
        out["norms"][k] = max(out["norms"].get(k, 0.0), d)

# ---- second experiment: the whole two-stage flow (rounds 0..2) with and without the 1e-7
# perturbation: how far do t (threshold counts), prototypes and the tagging similarities move?
import numpy as np                                                     # noqa: E402


def run_flow(pert):
    torch.set_num_threads(8)
    glob = oracle_net(C, g["init_seed"])
    if pert:
        with torch.no_grad():
            for p in glob.parameters():
                p.mul_(1.0 + pert)
    clients = [R.RefClient(args, i, data, g["users"][i], neg, [i]) for i in range(n_cl)]
    prototype, rec = None, {}
    for rnd in range(3):
        r = g["rounds"][rnd]
        w, protos = [], []
        for i, cl in enumerate(clients):
            net = copy.deepcopy(glob)
            if rnd < S1:
                ret = cl.stage1(net, r["train_orders"][i], with_proto=(rnd == S1 - 1),
                                negative_param=g["neg_lists"][i])
            else:
                ret = cl.stage2(rnd, net, prototype, g["neg_lists"][i], r["feat_orders"][i],
                                r["train_orders"][i])
                rec[f"sims_r{rnd}_c{i}"] = ret[5]
            w.append(copy.deepcopy(ret[0]))
            if len(ret) >= 5:
                protos.append(ret[4])
                rec[f"t_r{rnd}_c{i}"] = np.asarray(ret[3]) * N
                rec[f"proto_r{rnd}_c{i}"] = ret[4].numpy()
        glob.load_state_dict(R.fedavg(w, [N] * n_cl))
        if rnd >= S1 - 1:
            prototype = R.fedavg_proto(protos, [N] * n_cl, g["class_active_client_list"])
    return rec


a, b = run_flow(0.0), run_flow(1e-7)
flow = {"t_count_dev": 0.0, "proto_rel_to_max_dev": 0.0, "sim_abs_dev": 0.0}
for k in a:
    if k.startswith("t_"):
        flow["t_count_dev"] = max(flow["t_count_dev"], float(np.abs(a[k] - b[k]).max()))
    elif k.startswith("proto_"):
        flow["proto_rel_to_max_dev"] = max(flow["proto_rel_to_max_dev"],
                                           float(np.nanmax(np.abs(a[k] - b[k])) / np.nanmax(np.abs(a[k]))))
    else:
        for s1, s2 in zip(a[k], b[k]):
            s1, s2 = np.array(s1), np.array(s2)
            if np.isfinite(s1).all():
                flow["sim_abs_dev"] = max(flow["sim_abs_dev"], float(np.abs(s1 - s2).max()))
out["flow_1e-7"] = flow
print("flow:", flow)
json.dump(out, open(os.path.join(HERE, "conditioning.json"), "w"), indent=1)
worst = sorted(out["norms"].items(), key=lambda kv: -kv[1])[:8]
print("loss:", out["loss"]); print("worst:", worst)
