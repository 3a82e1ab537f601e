#!/bin/bash
# In-kernel clock and matrix-pipe duty of the conv-forward GEMM of one ResNet-18 layer, product forms 6 and 0 (fp32 pipe), from one
# rocprofv3 --pmc pass each (kernel-trace only) over tools/probe_conv.py at 1024 images (launches of 1-2 ms: GRBM_GUI_ACTIVE / 8 /
# duration is the clock within a few per cent on dispatches that long).    usage: tools/mfma_busy.sh [conv index] -> gpurun_out/mfma_busy.txt
CI=${1:-16}
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
: > gpurun_out/mfma_busy.txt
for sp in 6; do
  O=gpurun_out/mfma_busy_$sp; rm -rf $O; mkdir -p $O
  FM_MFMA_SPLIT=$sp rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/prof -- python3 tools/probe_conv.py 1024 $CI 0 > $O/log.txt 2>&1
  python3 - $O $sp >> gpurun_out/mfma_busy.txt <<'PY'
import csv, sys, glob, collections
O, sp = sys.argv[1], sys.argv[2]
f = glob.glob(O + '/prof/**/*counter_collection.csv', recursive=True)[0]
t = glob.glob(O + '/prof/**/*kernel_trace.csv', recursive=True)[0]
dur = {r['Dispatch_Id']: (int(r['End_Timestamp']) - int(r['Start_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(t))}
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    if 'igemm' in r['Kernel_Name'] or 'pconv' in r['Kernel_Name']:
        agg[r['Dispatch_Id']][r['Counter_Name']] += float(r['Counter_Value'])
for d, c in list(agg.items())[-3:]:
    ns, name = dur[d]
    cyc = c['GRBM_GUI_ACTIVE'] / 8
    print(f"FM_MFMA_SPLIT={sp} {name[:64]}  {ns / 1e3:.1f} us  clock {cyc / ns:.3f} GHz  matrix pipe busy {c['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024):.3f}")
PY
  rm -rf $O
done
cat gpurun_out/mfma_busy.txt
