#!/bin/bash
# GPU pass of round 5: the whole GPU suite, then the headline bench (both planes arms)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
mkdir -p gpurun_out
timeout 2400 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -25 > gpurun_out/r5_tests.txt
timeout 600 python3 bench.py --no-legs --sustain-s 0 > gpurun_out/r5_bench_planes.txt 2>&1
FM_PLANES=0 timeout 600 python3 bench.py --no-legs --sustain-s 0 --no-cpu-baseline > gpurun_out/r5_bench_noplanes.txt 2>&1
tail -5 gpurun_out/r5_tests.txt; tail -2 gpurun_out/r5_bench_planes.txt; tail -2 gpurun_out/r5_bench_noplanes.txt
