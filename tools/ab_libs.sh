#!/bin/bash
# same-box A/B of tuning builds (TUNE_TAG=<tag> tools/build_tuning.sh): tools/ab_libs.sh tag1 tag2 ...  (two-stream and one-stream step)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
export TMPDIR=/tmp
P=$PWD
for r in 1 2 3; do for t in "$@"; do
  echo -n "$t two-stream "; FEDMLP_HIP_LIB=$P/tune/libfedmlp_hip_tune_$t.so timeout 200 python3 bench.py --steps 40 --warmup 5 --no-legs --no-cpu-baseline --sustain-s 0 --no-profile 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'])"
  echo -n "$t one-stream "; FEDMLP_HIP_LIB=$P/tune/libfedmlp_hip_tune_$t.so timeout 200 python3 bench.py --steps 40 --warmup 5 --no-legs --no-cpu-baseline --sustain-s 0 --no-profile --one-stream 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'])"
done; done
