#!/bin/bash
# same-box A/B of a timing or variant build of the library against the shipped one: tools/gpu_ab.sh <lib.so> [probe args]
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
LIB=$1; shift
ARGS=${@:-256 8,13,18 0,1}
for r in 1 2; do
for l in fedmlp_amd/libfedmlp_hip.so $LIB; do
  echo "== $l"
  FEDMLP_HIP_LIB=$PWD/$l FM_DEBUG_REUSE_PLANES=1 timeout 200 python3 tools/probe_conv.py $ARGS 2>&1 | grep -v amdgpu.ids
done; done
