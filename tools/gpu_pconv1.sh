#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
mkdir -p gpurun_out
timeout 900 python3 -m pytest tests/test_kernels_gpu.py -x -q 2>&1 | tail -15 > gpurun_out/pconv1_tests.txt
FM_DEBUG_REUSE_PLANES=1 timeout 300 python3 tools/probe_conv.py 256 1,3,6,8,11,13,16,18 0,1 > gpurun_out/pconv1_probe.txt 2>&1
FM_PLANES=0 timeout 300 python3 tools/probe_conv.py 256 1,3,6,8,11,13,16,18 0,1 > gpurun_out/pconv1_probe_base.txt 2>&1
cat gpurun_out/pconv1_tests.txt gpurun_out/pconv1_probe.txt gpurun_out/pconv1_probe_base.txt
