#!/bin/bash
# round-6: row-major steps in the 64-row pconv tiles and in pwgrad.  Parity, A/B of the arms, overlap
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/r6g; mkdir -p $OUT
timeout 1200 python3 -m pytest tests/test_kernels_gpu.py tests/test_engine_gpu.py -q 2>&1 | tail -30 > $OUT/tests.txt; tail -5 $OUT/tests.txt
P=$PWD
probe() { FEDMLP_HIP_LIB=$P/$1 FM_DEBUG_REUSE_PLANES=1 timeout 200 python3 tools/probe_conv.py 256 2,5,7,13,18 0,1,2 2>&1 | grep -v amdgpu.ids; }
step() { FEDMLP_HIP_LIB=$P/$1 timeout 200 python3 bench.py --steps 40 --warmup 5 --no-legs --no-cpu-baseline --sustain-s 0 --no-profile $2 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], 'value', d['value'])"; }
for r in 1 2; do
  for l in tune/libfedmlp_hip_tune_base.so fedmlp_amd/libfedmlp_hip.so; do echo "== $l"; probe $l; done
done > $OUT/probe.txt 2>&1
for r in 1 2 3; do
  for l in tune/libfedmlp_hip_tune_base.so tune/libfedmlp_hip_tune_pc2.so tune/libfedmlp_hip_tune_pw.so fedmlp_amd/libfedmlp_hip.so; do echo "== $l two-stream"; step $l; done
  for l in tune/libfedmlp_hip_tune_base.so fedmlp_amd/libfedmlp_hip.so; do echo "== $l one-stream"; step $l --one-stream; done
done > $OUT/step.txt 2>&1
cat $OUT/probe.txt $OUT/step.txt
bash tools/gpu_overlap.sh > $OUT/overlap.txt 2>&1; tail -12 $OUT/overlap.txt
