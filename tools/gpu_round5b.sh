#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
mkdir -p gpurun_out
timeout 2400 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -25 > gpurun_out/r5_tests.txt
bash tools/gpu_prof3.sh > gpurun_out/r5_prof3.txt 2>&1
tail -8 gpurun_out/r5_tests.txt; cat gpurun_out/r5_prof3.txt
