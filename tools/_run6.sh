export TMPDIR=/tmp
OUT=gpurun_out/pmc; mkdir -p $OUT
CNT="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAVES"
rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d $OUT/ebf.sq -- python3 bench.py --model Efficient_b0 --precision bf16 --batch 512 --one-stream --steps 3 --warmup 1 --no-cpu-baseline --no-profile > $OUT/ebf.sq.log 2>&1
F=$(find $OUT/ebf.sq -name '*counter_collection.csv' | head -1)
python3 - "$F" > gpurun_out/ebf_sq.txt <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter(); seen=set()
meta={}
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]; agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    key=(r["Dispatch_Id"],k)
    if key not in seen:
        seen.add(key); calls[k]+=1
        meta[k]=(r.get("VGPR_Count"), r.get("Accum_VGPR_Count"), r.get("LDS_Block_Size"), r.get("Workgroup_Size"), r.get("Grid_Size"))
rows = sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", 0))[:45]
for k, c in rows:
    wc = c.get("SQ_WAVE_CYCLES", 1) or 1
    print(k[:100])
    print("   calls %d busy %.3g wave_cycles %.3g waves %.3g | parked %.3f issue_stall %.3f active %.3f (valu %.3f vmem %.3f) | meta %s"
          % (calls[k], c["SQ_BUSY_CYCLES"], wc, c["SQ_WAVES"], c["SQ_WAIT_ANY"]/wc, c["SQ_WAIT_INST_ANY"]/wc, c["SQ_ACTIVE_INST_ANY"]/wc,
             c["SQ_ACTIVE_INST_VALU"]/wc, c["SQ_ACTIVE_INST_VMEM"]/wc, meta[k]))
PY
rm -rf $OUT/ebf.sq
head -60 gpurun_out/ebf_sq.txt
