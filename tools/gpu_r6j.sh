#!/bin/bash
# round-6, VERDICT r5 item 6: the full-size shared-mask step against float64, per tensor, under each product form (FM_MFMA_SPLIT=0:
# fp32 matrix pipe; 9: nine partial products) and with the six-product form in the fp32-operand kernels (FM_PLANES=0)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/r6j; mkdir -p $OUT
T=tests/test_golden_r4_gpu.py::test_stage1_step_at_the_benchmarked_size_against_the_oracle_with_shared_relu_masks
for env in "" "FM_MFMA_SPLIT=0" "FM_MFMA_SPLIT=9" "FM_PLANES=0"; do
  env $env timeout 900 python3 -m pytest $T -q 2>&1 | tail -2
done
cp gpurun_out/parity_step_full_shared_masks*.json $OUT/
