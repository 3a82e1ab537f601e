#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
mkdir -p gpurun_out
timeout 300 python3 -m pytest tests/test_kernels_gpu.py -x -q 2>&1 | tail -15 > gpurun_out/exp_tests.txt
cat gpurun_out/exp_tests.txt
for r in 1 0; do
echo "== FM_PWGRAD_RING=$r"
FM_PWGRAD_RING=$r FM_DEBUG_REUSE_PLANES=1 timeout 200 python3 tools/probe_conv.py 256 1,3,6,8 2 2>&1 | grep -v amdgpu.ids
done
echo "== FM_PWGRAD_RING_MINW=7"
FM_PWGRAD_RING_MINW=7 FM_DEBUG_REUSE_PLANES=1 timeout 200 python3 tools/probe_conv.py 256 11,13,16,18 2 2>&1 | grep -v amdgpu.ids
FM_PWGRAD_RING=0 FM_DEBUG_REUSE_PLANES=1 timeout 200 python3 tools/probe_conv.py 256 11,13,16,18 2 2>&1 | grep -v amdgpu.ids
