#!/bin/bash
# Counter groups (one rocprofv3 --pmc pass each, kernel-trace only, own time limit) over ONE short one-stream bench run, summed
# per kernel name -> gpurun_out/pmc_groups/<tag>.<group>.txt     usage: tools/pmc_groups.sh <tag> <kernel name filter> <bench args...>
TAG=$1; FILT=$2; shift 2
OUT=gpurun_out/pmc_groups; mkdir -p $OUT
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
export TMPDIR=/tmp
declare -A G
G[sq]="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA"
G[ta]="TA_TA_BUSY TA_BUFFER_TOTAL_CYCLES TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES"
G[tcp]="TCP_PENDING_STALL_CYCLES TCP_TCP_TA_DATA_STALL_CYCLES TCP_TCR_TCP_STALL_CYCLES TCP_GATE_EN1"
G[td]="TD_TD_BUSY TD_TC_STALL GRBM_GUI_ACTIVE"
for g in ${PMC_GROUPS:-sq tcp td}; do
  rm -rf $OUT/$TAG.$g
  timeout 240 rocprofv3 --pmc ${G[$g]} --kernel-trace --output-format csv -d $OUT/$TAG.$g -- python3 bench.py "$@" --one-stream --steps 1 --warmup 1 --no-cpu-baseline --no-profile > $OUT/$TAG.$g.log 2>&1
  echo "group $g rc=$?"
  F=$(find $OUT/$TAG.$g -name '*counter_collection.csv' | head -1)
  [ -n "$F" ] && python3 - "$F" "$FILT" > $OUT/$TAG.$g.txt <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter(); seen = set()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if sys.argv[2] not in k: continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (r["Dispatch_Id"], k)
    if key not in seen: seen.add(key); calls[k] += 1
for k, c in sorted(agg.items(), key=lambda kv: -sum(kv[1].values()))[:24]:
    print(k[:100], "calls", calls[k])
    print("    " + "  ".join(f"{n}={v:.4g}" for n, v in sorted(c.items())))
PY
  rm -rf $OUT/$TAG.$g
  cat $OUT/$TAG.$g.txt 2>/dev/null | head -60
done
