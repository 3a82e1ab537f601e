#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/tl; rm -rf gpurun_out/tl/prof
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl/prof -- python3 bench.py --steps 12 --warmup 4 --no-legs --sustain-s 0 --no-cpu-baseline --no-profile > gpurun_out/tl/bench.txt 2>&1
f=$(find gpurun_out/tl/prof -name '*kernel_trace.csv' | head -1)
head -2 "$f" | cut -c1-600
python3 tools/timeline.py "$f" "pconv_kernel|pwgrad|igemm_kernel|wgrad_kernel|stem_rows" > gpurun_out/tl/timeline.txt 2>&1
python3 tools/overlap.py "$f" > gpurun_out/tl/overlap.txt 2>&1
rm -rf gpurun_out/tl/prof
cat gpurun_out/tl/timeline.txt gpurun_out/tl/overlap.txt; tail -c 200 gpurun_out/tl/bench.txt
