#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
mkdir -p gpurun_out
timeout 300 python3 -m pytest tests/test_kernels_gpu.py -x -q 2>&1 | tail -15 > gpurun_out/exp_tests.txt
cat gpurun_out/exp_tests.txt
FM_DEBUG_REUSE_PLANES=1 timeout 200 python3 tools/probe_conv.py 256 1,3,6,8,11,13,16,18 0,1 2>&1 | grep -v amdgpu.ids
timeout 300 python3 bench.py --no-legs --sustain-s 0 --no-cpu-baseline --steps 40 2>&1 | tail -1 | cut -c1-330
