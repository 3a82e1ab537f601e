#!/bin/bash
# same-box A/B of two builds of the library (build/ab/lib_a.so vs build/ab/lib_b.so): alternating bench runs.
# usage: tools/ab.sh <reps> <bench args...>
REPS=$1; shift
cp fedmlp_amd/libfedmlp_hip.so /tmp/lib_keep.so
for rep in $(seq $REPS); do for v in a b; do
  cp build/ab/lib_$v.so fedmlp_amd/libfedmlp_hip.so
  python bench.py "$@" --no-cpu-baseline --no-profile 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lib_$v', d['ms_per_step'], 'ms', d['value'], 'img/s')"
done; done
cp /tmp/lib_keep.so fedmlp_amd/libfedmlp_hip.so
