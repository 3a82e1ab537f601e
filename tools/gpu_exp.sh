#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
mkdir -p gpurun_out
for v in 1 0 1 0; do
echo "== FM_PCONV_NW4=$v (tuning build)"
FEDMLP_HIP_LIB=$PWD/fedmlp_amd/libfedmlp_hip_tune.so FM_PCONV_NW4=$v timeout 300 python3 bench.py --no-legs --sustain-s 0 --no-cpu-baseline --steps 60 2>&1 | tail -1 | cut -c88-200
done
