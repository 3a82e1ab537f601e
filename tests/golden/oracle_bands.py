#!/usr/bin/env python3
"""How far does the REFERENCE move from itself when only the fp32 summation order changes?

The free-running two-stage replays (tests/test_golden_r2_gpu.py) and the full-size `train` step (tests/test_golden_r3_gpu.py)
compare the engine with goldens the reference's trainer produced in ONE arithmetic (oneDNN, 8 threads).  Their bounds (the
similarity band `delta` of the pick check, the worst-tensor gradient bound) have to cover what a different summation order alone
does to such a run -- a property of the experiment, not of the engine.  This script measures it: it re-runs
tests/golden/make_golden.py (which imports /root/reference) for `fedmlp64` and for the `train` record of `step_full_variants` under
other orders (3 threads: the BatchNorm / weight-gradient reductions split differently; oneDNN off: torch's native convolution
kernels) and compares each re-run with the committed golden.  Output: tests/golden/oracle_bands.json (committed; the tests derive
their bounds from it).  Runs only in the build container.

usage: python tests/golden/oracle_bands.py [fedmlp64] [train]"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
VARIANTS = {"threads3": {"GOLDEN_THREADS": "3"}, "mkldnn_off": {"GOLDEN_MKLDNN": "0"}}


def run_variant(which, env_extra, out):
    env = dict(os.environ, GOLDEN_OUT=out, **env_extra)
    subprocess.run([sys.executable, os.path.join(HERE, "make_golden.py"), which], check=True, env=env, cwd=ROOT,
                   stdout=subprocess.DEVNULL)


def fedmlp64_band(ref_dir, new_dir):
    """similarity rows, picks, losses and final norms of two runs of the same two-stage flow"""
    a = json.load(open(os.path.join(ref_dir, "traj_fedmlp64.json")))
    b = json.load(open(os.path.join(new_dir, "traj_fedmlp64.json")))
    Pa = np.load(os.path.join(ref_dir, "traj_fedmlp64_protos.npz"))
    Pb = np.load(os.path.join(new_dir, "traj_fedmlp64_protos.npz"))
    S1 = a["S1"]
    rep = {"sim_first_round": 0.0, "sim_later_rounds": 0.0, "picks_differing": 0, "picks_total": 0, "loss": 0.0, "norms": 0.0,
           "bn_bias_norms": 0.0, "proto": 0.0}
    for key in Pa.files:
        if key.endswith("_sim"):
            rnd = int(key.split("_")[0][1:])
            sa, sb = Pa[key].astype(np.float64), Pb[key].astype(np.float64)
            pa, pb = Pa[key[:-4] + "_pool"], Pb[key[:-4] + "_pool"]
            if np.isnan(sa).all():
                continue
            rng = float(np.nanmax(sa) - np.nanmin(sa))
            # compare on the samples both pools hold (later rounds: the pools themselves may differ by earlier picks)
            ia = {int(v): j for j, v in enumerate(pa)}
            common = [(ia[int(v)], j) for j, v in enumerate(pb) if int(v) in ia]
            d = max(abs(sa[i] - sb[j]) for i, j in common) / rng if common else 0.0
            k = "sim_first_round" if rnd == S1 else "sim_later_rounds"
            rep[k] = max(rep[k], float(d))
        elif key.endswith("_proto"):
            pa_, pb_ = Pa[key].astype(np.float64), Pb[key].astype(np.float64)
            ok = ~np.isnan(pa_)
            rep["proto"] = max(rep["proto"], float(np.abs(pa_[ok] - pb_[ok]).max() / (np.abs(pa_[ok]).max() + 1e-30)))
    for ra, rb in zip(a["rounds"], b["rounds"]):
        for la, lb in zip(ra["loss"], rb["loss"]):
            rep["loss"] = max(rep["loss"], abs(la - lb) / abs(la))
        for na, nb in zip(ra["norms"], rb["norms"]):
            for k in na:
                d = abs(na[k] - nb[k]) / (abs(na[k]) + 1e-30)
                if k.endswith("bn1.bias") or k.endswith("bn2.bias") or k.endswith("downsample.1.bias"):
                    rep["bn_bias_norms"] = max(rep["bn_bias_norms"], d)
                elif "num_batches" not in k:
                    rep["norms"] = max(rep["norms"], d)
        for ta, tb in zip(ra.get("traindata_idx", []), rb.get("traindata_idx", [])):
            for la, lb in zip(ta, tb):
                rep["picks_total"] += len(la)
                rep["picks_differing"] += len(set(la) ^ set(lb))
    return rep


def train_band(ref_dir, new_dir):
    """per-tensor gradient records of the full-size `train` step, with tests/helpers.grad_errors' measure on the records"""
    a = json.load(open(os.path.join(ref_dir, "step_full_variants.json")))["train"]
    b = json.load(open(os.path.join(new_dir, "step_full_variants.json")))["train"]
    typ = float(np.median([v["absmax"] for v in a["grads"].values()]))
    worst, allv = ("", 0.0), []
    for k, w in a["grads"].items():
        g = b["grads"][k]
        e = abs(g["norm"] - w["norm"]) / (w["norm"] + 1e-30)
        e = max(e, max(abs(x - y) for x, y in zip(g["head"], w["head"])) / max(w["absmax"], 1e-3 * typ))
        allv.append(e)
        if e > worst[1]:
            worst = (k, e)
    return {"loss": abs(a["loss"] - b["loss"]) / abs(a["loss"]), "worst_tensor": worst[0], "worst": worst[1],
            "median": float(np.median(allv))}


def main():
    which = sys.argv[1:] or ["fedmlp64", "train"]
    path = os.path.join(HERE, "oracle_bands.json")
    out = json.load(open(path)) if os.path.exists(path) else {}
    for name, env in VARIANTS.items():
        with tempfile.TemporaryDirectory() as tmp:
            if "fedmlp64" in which:
                run_variant("fedmlp64", env, tmp)
                out.setdefault("traj_fedmlp64", {})[name] = fedmlp64_band(HERE, tmp)
                print("traj_fedmlp64", name, out["traj_fedmlp64"][name], flush=True)
            if "train" in which:
                run_variant("step_full_variants", dict(env, GOLDEN_ONLY_TRAIN="1"), tmp)
                out.setdefault("step_full_train", {})[name] = train_band(HERE, tmp)
                print("step_full_train", name, out["step_full_train"][name], flush=True)
            json.dump(out, open(path, "w"), indent=1)


if __name__ == "__main__":
    main()
