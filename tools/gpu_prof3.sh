#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
bash tools/prof_stats.sh r5_planes --steps 20 --warmup 3 --no-legs --sustain-s 0 > /dev/null 2>&1
for t in r5_planes; do echo "== $t"; python3 - $t <<'PY'
import csv,sys
rows=list(csv.DictReader(open(f"gpurun_out/{sys.argv[1]}/kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total ms/step", tot/1e6/23)
for r in rows[:16]:
    print(f'{r["Name"][:70]:70s} calls {int(r["Calls"]):5d} avg {float(r["AverageNs"])/1e3:8.1f} us  per-step {float(r["TotalDurationNs"])/1e6/23:6.3f} ms')
PY
tail -c 300 gpurun_out/$t/bench.json; echo; done
timeout 300 python3 bench.py --no-legs --sustain-s 0 --no-cpu-baseline --steps 40 2>&1 | tail -1 | cut -c1-400
