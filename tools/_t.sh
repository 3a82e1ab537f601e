#!/bin/bash
cd $GRAFT_REPO_ROOT
cp fedmlp_amd/libfedmlp_hip.so /tmp/keep.so
cp build/ab/lib_tune.so fedmlp_amd/libfedmlp_hip.so
timeout 600 python tools/knob_diff.py FM_PW_PROJ_BWD 224 32 bf16 2>&1 | grep -v amdgpu.ids
cp /tmp/keep.so fedmlp_amd/libfedmlp_hip.so
bash tools/ab.sh 3 --model Efficient_b0 --precision bf16 --batch 512
cp build/ab/lib_a.so fedmlp_amd/libfedmlp_hip.so
python tools/op_profile.py --precision bf16 --streams 1 2>/dev/null | grep -E "total|proj_bwd|k_se_bwd/|bnact_bwd/|proj_wgrad/|proj_dgrad/" | head -40
