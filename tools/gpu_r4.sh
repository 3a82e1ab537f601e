#!/bin/bash
# round-4 GPU pass: full test suite + the ONE bench line (headline + sustained repeat + every leg) -> gpurun_out/<tag>/
# usage: tools/gpu_r4.sh <tag> [pytest args]
TAG=${1:-r4}; shift
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
OUT=gpurun_out/$TAG; mkdir -p $OUT
python -m pytest tests -m gpu -q "$@" 2>&1 | tail -60 > $OUT/pytest.log; tail -8 $OUT/pytest.log
cp gpurun_out/parity_*.json $OUT/ 2>/dev/null
python bench.py > $OUT/bench.json 2> $OUT/bench_err.log; tail -3 $OUT/bench_err.log
python - "$OUT/bench.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d.get("roofline") or {}
print("headline", d["value"], "img/s", d["ms_per_step"], "ms", r.get("frac"), r.get("whole_step_frac"), "sustained", d.get("sustained_ms_per_step"),
      "cpu", (d.get("cpu_baseline") or {}).get("value"))
for k, v in (d.get("legs") or {}).items():
    if "error" in v:
        print(k, "ERROR", v["error"]); continue
    rr = v.get("roofline") or {}
    print(k, v.get("value"), v.get("unit"), v.get("ms_per_step", v.get("ms_per_pass", v.get("ms_per_call", v.get("ms_per_aggregation")))), "ms",
          rr.get("bound"), rr.get("frac"), rr.get("whole_step_frac"), "cpu", (v.get("cpu_baseline") or {}).get("value"), "wall", v.get("leg_wall_s"))
PY
