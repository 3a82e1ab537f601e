#!/bin/bash
# builds nothing: runs the prebuilt probe (build/pconv_probe) and the shipped kernels' per-layer timings on the same box
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
mkdir -p gpurun_out
timeout 300 ./build/pconv_probe > gpurun_out/pconv_probe.txt 2>&1

cat gpurun_out/pconv_probe.txt
