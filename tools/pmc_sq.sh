#!/bin/bash
# One rocprofv3 --pmc pass of SQ counters (kernel-trace only) over a short bench.py run, summed per kernel name.
# usage: tools/pmc_sq.sh <tag> <bench args...>      -> gpurun_out/pmc/<tag>.sq.txt
TAG=$1; shift
OUT=gpurun_out/pmc
mkdir -p $OUT
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
export TMPDIR=/tmp
CNT="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES"
rm -rf $OUT/$TAG.sq
rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d $OUT/$TAG.sq -- python3 bench.py "$@" --steps 4 --warmup 2 --no-cpu-baseline --no-profile > $OUT/$TAG.sq.log 2>&1
F=$(find $OUT/$TAG.sq -name '*counter_collection.csv' | head -1)
python3 - "$F" > $OUT/$TAG.sq.txt <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
seen = set()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]; agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (r["Dispatch_Id"], k)
    if key not in seen: seen.add(key); calls[k] += 1
rows = sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:14]
for k, c in rows:
    wc = c.get("SQ_WAVE_CYCLES", 1) or 1
    print(k[:70])
    print("   calls %d  wave_cycles %.3g | parked %.3f  issue_stall %.3f (lds %.3f)  active %.3f | lds_conflict/lds_active %.3f | mfma_busy %.3g"
          % (calls[k], wc, c["SQ_WAIT_ANY"] / wc, c["SQ_WAIT_INST_ANY"] / wc, c["SQ_WAIT_INST_LDS"] / wc, c["SQ_ACTIVE_INST_ANY"] / wc,
             c["SQ_LDS_BANK_CONFLICT"] / max(c["SQ_LDS_IDX_ACTIVE"], 1), c["SQ_VALU_MFMA_BUSY_CYCLES"]))
PY
rm -rf $OUT/$TAG.sq
cat $OUT/$TAG.sq.txt
