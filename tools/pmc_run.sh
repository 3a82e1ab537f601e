#!/bin/bash
# Two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; kernel-trace only, never combined with other
# tracing) of a short bench.py run -> gpurun_out/pmc/<tag>.json (copy to profiles/rNN/pmc_traffic.json)
# usage: tools/pmc_run.sh <tag> <workload key> <steps> <warmup> <bench args...>
TAG=$1; KEY=$2; STEPS=$3; WARM=$4; shift 4
OUT=gpurun_out/pmc
mkdir -p $OUT
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $OUT/$TAG.$C
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/$TAG.$C -- python3 bench.py "$@" --one-stream --steps $STEPS --warmup $WARM --no-cpu-baseline --no-profile > $OUT/$TAG.$C.log 2>&1
done
F=$(find $OUT/$TAG.FETCH_SIZE -name '*counter_collection.csv' | head -1)
W=$(find $OUT/$TAG.WRITE_SIZE -name '*counter_collection.csv' | head -1)
python3 tools/pmc_traffic.py "$F" "$W" $OUT/pmc_traffic.json "$KEY" $((STEPS+WARM)) "bench.py $* --one-stream --steps $STEPS --warmup $WARM"
rm -rf $OUT/$TAG.FETCH_SIZE $OUT/$TAG.WRITE_SIZE
python3 - <<PY
import json
d=json.load(open("$OUT/pmc_traffic.json"))["$KEY"]["kernels"]
for k in list(d)[:6]+["whole_step"]:
    print(k[:90], d[k]["hbm_bytes_per_launch_corrected"])
PY
