export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/pwc; mkdir -p $OUT
run() { # name counters...
  n=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$n -- python3 tools/pw_time.py --imgs 1024 --reps 2 --only 1,2,5,4 > $OUT/$n.log 2>&1
  f=$(find $OUT/$n -name '*counter_collection.csv' | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter(); seen=set()
for r in csv.DictReader(open(sys.argv[1])):
    k=r["Kernel_Name"]
    if "pw_conv" not in k: continue
    key=(k, r["Grid_Size"])
    agg[key][r["Counter_Name"]]+=float(r["Counter_Value"])
    if (r["Dispatch_Id"],) not in seen: seen.add((r["Dispatch_Id"],)); n[key]+=1
for key,c in agg.items():
    print(key[0][-40:], 'grid', key[1], 'calls', n[key], {k: f"{v/n[key]:.4g}" for k,v in c.items()})
PY
  rm -rf $OUT/$n
}
run a TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
run b TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TA_BUSY_sum GRBM_GUI_ACTIVE
run c TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_BUSY_sum TCC_HIT_sum
cat $OUT/a.log | tail -8
