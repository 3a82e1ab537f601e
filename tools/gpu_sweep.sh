#!/bin/bash
# same-box sweep of one tuning knob of the tuning build over the two-stream step: tools/gpu_sweep.sh <ENV_NAME> <v1> <v2> ...
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
export TMPDIR=/tmp
K=$1; shift
P=$PWD
step() { FEDMLP_HIP_LIB=$P/tune/libfedmlp_hip_tune.so timeout 200 python3 bench.py --steps 40 --warmup 5 --no-legs --no-cpu-baseline --sustain-s 0 --no-profile 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'])"; }
for r in 1 2 3; do for v in "$@"; do echo -n "$K=$v  "; env $K=$v bash -c "$(declare -f step); P=$P; step"; done; done
