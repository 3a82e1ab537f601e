#!/usr/bin/env python3
"""Oracle half of the paired mAP study (tests/map_flow.py): runs in the build container, writes the committed fixture
tests/golden/map_study_oracle.json.   usage: python tools/map_study.py [n_seeds]   |   python tools/map_study.py converged  (the three converged-flow orders)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                                      # noqa: E402
from tests import map_flow as F                   # noqa: E402

torch.set_num_threads(8)
if len(sys.argv) > 1 and sys.argv[1] == "converged":
    # oracle half of tests/test_eval_gpu.py::test_map_after_converged_training_matches_oracle -> its committed fixture
    from tests import test_eval_gpu as E          # noqa: E402
    out = {"config": dict(E.CONVERGED), "orders": {}}
    for o in E.CONVERGED_ORDERS:
        t0 = time.time()
        out["orders"][str(o)] = E.map_flow_oracle(order_seed=o, **E.CONVERGED)
        print(f"order {o}: {out['orders'][str(o)]}  ({time.time() - t0:.0f} s)", flush=True)
        with open(os.path.join(ROOT, "tests", "golden", "map_converged_oracle.json"), "w") as f:
            json.dump(out, f, indent=1)
    sys.exit(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else len(F.SEEDS)
test = F.make_split(F.N_TEST, F.TEST_SEED, False)
out = {"config": {k: getattr(F, k) for k in ("C", "N_CL", "N_LOCAL", "HW", "BS", "S1", "ROUNDS", "LR", "N_TEST", "SIGNAL",
                                               "LABEL_NOISE")},
       "prevalence": float(test[2].mean()), "runs": {}}
path = os.path.join(ROOT, "tests", "golden", "map_study_oracle.json")
if os.path.exists(path):                          # resume: seeds already in the fixture are kept (same config only)
    old = json.load(open(path))
    if old.get("config") == out["config"]:
        out["runs"] = old["runs"]
for s in F.SEEDS[:n]:
    if str(s) in out["runs"]:
        continue
    t0 = time.time()
    m, a = F.run_oracle(s, test)
    out["runs"][str(s)] = {"mAP": m, "auc": a}
    print(f"seed {s}: mAP {m:.5f} AUROC {a:.5f}  ({time.time() - t0:.0f} s)", flush=True)
    out["mAP"] = F.summarise([r["mAP"] for r in out["runs"].values()])
    out["auc"] = F.summarise([r["auc"] for r in out["runs"].values()])
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
print(json.dumps({"mAP": out["mAP"], "auc": out["auc"]}))
