#!/bin/bash
# HIP half of the paired mAP study (tests/test_eval_gpu.py, seeds of tests/golden/map_study_oracle.json) in two arms:
# the shipped library defaults, and FM_STEM_PACKED=0 (the [7][8][4] stem layout: another summation order of the first conv).
# -> gpurun_out/parity_map.json (default arm, written by the test), gpurun_out/parity_map_arms.json (both arms side by side)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
mkdir -p gpurun_out
FM_STEM_PACKED=0 python -m pytest tests/test_eval_gpu.py -m gpu -q -k paired_study 2>&1 | tail -2
cp gpurun_out/parity_map.json gpurun_out/parity_map_stem_padded.json
python -m pytest tests/test_eval_gpu.py -m gpu -q -k paired_study 2>&1 | tail -2
python - <<'PY'
import json
a = json.load(open("gpurun_out/parity_map.json")); b = json.load(open("gpurun_out/parity_map_stem_padded.json"))
import numpy as np
def summ(v):
    v = np.asarray(v, float); return {"mean": float(v.mean()), "se": float(v.std(ddof=1) / np.sqrt(len(v))), "n": len(v)}
seeds = a["seeds"]
dm = [a["hip"]["runs"][str(s)]["mAP"] - b["hip"]["runs"][str(s)]["mAP"] for s in seeds]
da = [a["hip"]["runs"][str(s)]["auc"] - b["hip"]["runs"][str(s)]["auc"] for s in seeds]
out = {"seeds": seeds,
       "default_minus_oracle": a["paired_difference_hip_minus_oracle"],
       "stem_padded_minus_oracle": b["paired_difference_hip_minus_oracle"],
       "default_minus_stem_padded": {"mAP": summ(dm), "auc": summ(da)},
       "signs_default": {"mAP_below_oracle": int(sum(a["hip"]["runs"][str(s)]["mAP"] < a["oracle"]["runs"][str(s)]["mAP"] for s in seeds)),
                         "auc_below_oracle": int(sum(a["hip"]["runs"][str(s)]["auc"] < a["oracle"]["runs"][str(s)]["auc"] for s in seeds))},
       "signs_stem_padded": {"mAP_below_oracle": int(sum(b["hip"]["runs"][str(s)]["mAP"] < b["oracle"]["runs"][str(s)]["mAP"] for s in seeds)),
                             "auc_below_oracle": int(sum(b["hip"]["runs"][str(s)]["auc"] < b["oracle"]["runs"][str(s)]["auc"] for s in seeds))},
       "note": "the loss heads and the host-side evaluation already use IEEE expf / logf / division (csrc/heads.hip sigmoidf_, "
               "fedmlp_amd/evaluations.py): there is no fast-math arm to switch in the ResNet-18 fp32 flow this study runs"}
json.dump(out, open("gpurun_out/parity_map_arms.json", "w"), indent=1)
print(json.dumps({k: out[k] for k in ("default_minus_oracle", "stem_padded_minus_oracle", "default_minus_stem_padded", "signs_default", "signs_stem_padded")}))
PY
