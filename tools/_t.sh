#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_effnet_gpu.py -x -q -k "fused_project" 2>&1 | tail -5
cp fedmlp_amd/libfedmlp_hip.so /tmp/keep.so
cp build/ab/lib_tune.so fedmlp_amd/libfedmlp_hip.so
timeout 600 python tools/knob_diff.py FM_PW_PROJ_BWD_F32 224 16 fp32 2>&1 | grep -v amdgpu.ids
run() { python bench.py "$@" --no-cpu-baseline --no-profile 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], 'ms', d['value'], 'img/s')"; }
for rep in 1 2 3; do for v in 0 1; do
  echo -n "fp32 two-stream PROJ_F32=$v: "; FM_PW_PROJ_BWD_F32=$v run --model Efficient_b0 --batch 256
done; done
cp /tmp/keep.so fedmlp_amd/libfedmlp_hip.so
python tools/op_profile.py --precision fp32 --batch 256 --streams 1 2>/dev/null | grep -E "total|proj_bwd|k_se_bwd/|bnact_bwd/|proj_wgrad/|proj_dgrad/" | head -20
