#!/bin/bash
# round-6: merged BN finalize, stem backward (xhat from the pooled value, 2x2 apply).  Parity, then kernel statistics
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/r6e; mkdir -p $OUT
timeout 1500 python3 -m pytest tests/test_kernels_gpu.py tests/test_engine_gpu.py tests/test_golden_r4_gpu.py tests/test_golden_r2_gpu.py -q 2>&1 | tail -30 > $OUT/tests.txt; tail -5 $OUT/tests.txt
P=$PWD
step() { FEDMLP_HIP_LIB=$P/tune/libfedmlp_hip_tune.so timeout 200 python3 bench.py --steps 40 --warmup 5 --no-legs --no-cpu-baseline --sustain-s 0 --no-profile 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], 'value', d['value'])"; }
for r in 1 2; do
  for a in 0 1; do for b in 0 1; do
    echo "== XHAT_FROM_POOLED=$a APPLY_2X2=$b"; FM_STEM_XHAT_FROM_POOLED=$a FM_STEM_APPLY_2X2=$b step
  done; done
done > $OUT/step.txt 2>&1
cat $OUT/step.txt
timeout 300 bash tools/prof_stats.sh r6e_s1 --steps 30 --warmup 3 --no-legs --sustain-s 0 > $OUT/prof_s1.log 2>&1
cp gpurun_out/r6e_s1/kernel_stats.csv $OUT/kernel_stats_s1.csv
