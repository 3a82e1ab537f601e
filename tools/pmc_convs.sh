#!/bin/bash
# HBM-side traffic (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, kernel-trace only) of single conv launches per
# ResNet-18 layer: tools/pmc_convs.sh <tag> <imgs> <layers> <ops>   -> gpurun_out/pmc_convs_<tag>.txt
# (FETCH_SIZE doubled, KiB -> bytes: MI355X_MICROARCH.md, HBM section)
TAG=$1; IMGS=$2; LAYERS=$3; OPS=$4
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/pmc_convs
mkdir -p $OUT
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $OUT/$TAG.$C
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/$TAG.$C -- python3 tools/probe_conv.py $IMGS $LAYERS $OPS > $OUT/$TAG.$C.log 2>&1
done
python3 - <<PY > gpurun_out/pmc_convs_$TAG.txt
import csv, glob
def rows(c):
    f = glob.glob("$OUT/$TAG.%s/**/*counter_collection.csv" % c, recursive=True)[0]
    out = []
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == c and ("pconv_kernel" in r["Kernel_Name"] or "pwgrad" in r["Kernel_Name"] or "igemm" in r["Kernel_Name"]):
            out.append((int(r["Dispatch_Id"]), r["Kernel_Name"], float(r["Counter_Value"])))
    out.sort()
    return out
F, W = rows("FETCH_SIZE"), rows("WRITE_SIZE")
# consecutive launches of one probe line: 3 warm-up + 10 timed = 13
i = 0
lines = [l.strip() for l in open("$OUT/$TAG.FETCH_SIZE.log") if l.startswith("conv")]
k = 0
while i + 13 <= len(F):
    f = sum(x[2] for x in F[i:i+13]) / 13 * 2 * 1024
    w = sum(x[2] for x in W[i:i+13]) / 13 * 1024 if i + 13 <= len(W) else float("nan")
    name = F[i][1][:44]
    print(f"{lines[k] if k < len(lines) else '?':100s} | {name:44s} fetch {f/1e6:8.1f} MB  write {w/1e6:8.1f} MB")
    i += 13; k += 1
PY
cat gpurun_out/pmc_convs_$TAG.txt
rm -rf $OUT
