#!/bin/bash
# round-6: weight gradients enter the side stream behind their conv's data gradient (FM_WGRAD_LAG).  Parity, A/B, overlap
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/r6h; mkdir -p $OUT
timeout 1200 python3 -m pytest tests/test_engine_gpu.py tests/test_golden_r4_gpu.py -q 2>&1 | tail -30 > $OUT/tests.txt; tail -5 $OUT/tests.txt
P=$PWD
step() { FEDMLP_HIP_LIB=$P/tune/libfedmlp_hip_tune.so timeout 200 python3 bench.py --steps 40 --warmup 5 --no-legs --no-cpu-baseline --sustain-s 0 --no-profile $1 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], 'value', d['value'])"; }
for r in 1 2 3; do
  for lag in 0 1; do echo "== LAG=$lag stage1"; FM_WGRAD_LAG=$lag step; echo "== LAG=$lag train"; FM_WGRAD_LAG=$lag step "--workload train"; done
done > $OUT/step.txt 2>&1
cat $OUT/step.txt
bash tools/gpu_overlap.sh > $OUT/overlap.txt 2>&1; grep -v "simple_timer\|^\"K" $OUT/overlap.txt | tail -40
