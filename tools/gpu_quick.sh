#!/bin/bash
# short, individually bounded commands: a hung kernel must not eat the GPU budget
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
mkdir -p gpurun_out
timeout 240 python3 -m pytest tests/test_kernels_gpu.py -x -q 2>&1 | tail -4 > gpurun_out/quick_tests.txt
if grep -q passed gpurun_out/quick_tests.txt && ! grep -q failed gpurun_out/quick_tests.txt; then
  timeout 400 python3 -m pytest tests/test_engine_gpu.py tests/test_golden_r4_gpu.py -x -q 2>&1 | tail -4 >> gpurun_out/quick_tests.txt
  timeout 300 bash tools/gpu_prof3.sh > gpurun_out/r5_prof3.txt 2>&1
fi
cat gpurun_out/quick_tests.txt gpurun_out/r5_prof3.txt
