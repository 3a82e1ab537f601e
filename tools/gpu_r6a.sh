#!/bin/bash
# round-6 first pass: kernel parity, same-box A/B of the r5 library against the rebuilt one (waits tied to the accumulators) and
# the FM_PCONV_FLAGS arms of the tuning build, then the step time of each.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/r6a; mkdir -p $OUT
timeout 600 python3 -m pytest tests/test_kernels_gpu.py tests/test_engine_gpu.py -x -q 2>&1 | tail -6 > $OUT/kernels_tests.txt; cat $OUT/kernels_tests.txt
P=$PWD
probe() { FEDMLP_HIP_LIB=$P/$1 FM_DEBUG_REUSE_PLANES=1 timeout 200 python3 tools/probe_conv.py 256 1,5,7,8,13,18 0,1,2 2>&1 | grep -v amdgpu.ids; }
step() { FEDMLP_HIP_LIB=$P/$1 timeout 200 python3 bench.py --steps 40 --warmup 5 --no-legs --no-cpu-baseline --sustain-s 0 --no-profile 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], 'value', d['value'])"; }
for r in 1 2; do
  echo "== r5 lib"; probe tune/libfedmlp_hip_r5.so
  echo "== new lib"; probe fedmlp_amd/libfedmlp_hip.so
  for f in 1 2 3 4; do echo "== tune FM_PCONV_FLAGS=$f"; FM_PCONV_FLAGS=$f probe tune/libfedmlp_hip_tune.so; done
done > $OUT/probe.txt 2>&1
for r in 1 2; do
  echo "== r5 lib"; step tune/libfedmlp_hip_r5.so
  echo "== new lib"; step fedmlp_amd/libfedmlp_hip.so
  for f in 1 2 3 4; do echo "== tune FM_PCONV_FLAGS=$f"; FM_PCONV_FLAGS=$f step tune/libfedmlp_hip_tune.so; done
done > $OUT/step.txt 2>&1
cat $OUT/step.txt
