#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { python bench.py "$@" --no-cpu-baseline --no-profile 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], 'ms', d['value'], 'img/s')"; }
for rep in 1 2 3; do for v in 0 1; do
  echo -n "bf16 two-stream PROJ_BWD=$v: "; FM_PW_PROJ_BWD=$v run --model Efficient_b0 --precision bf16 --batch 512
done; done
for v in 0 1; do
  echo -n "bf16 one-stream PROJ_BWD=$v: "; FM_PW_PROJ_BWD=$v run --model Efficient_b0 --precision bf16 --batch 512 --one-stream
done
FM_PW_PROJ_BWD=1 python tools/op_profile.py --precision bf16 --streams 1 2>/dev/null | grep -E "total|@40[0-3] |proj_bwd|k_se_bwd/|bnact_bwd/|proj_wgrad/|proj_dgrad/" | head -40
