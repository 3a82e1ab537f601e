#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
mkdir -p gpurun_out
for n in 0 32; do
  lib=build/exp/lib_x$n.so; [ $n = 0 ] && lib=fedmlp_amd/libfedmlp_hip.so
  echo "== no tap selects / masks = $n"
  FEDMLP_HIP_LIB=$PWD/$lib FM_DEBUG_REUSE_PLANES=1 timeout 120 python3 tools/probe_conv.py 256 3,8,13,18 0 2>&1 | grep -v amdgpu.ids
done
FM_MFMA_SPLIT=9 timeout 600 python3 -m pytest tests/test_kernels_gpu.py tests/test_engine_gpu.py -q 2>&1 | grep -E "^FAILED|^E  " | cut -c1-220 | head -30 > gpurun_out/nine_failures.txt
cat gpurun_out/nine_failures.txt
