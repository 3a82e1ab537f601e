#!/bin/bash
# bench one workload under a list of environment settings: env_sweep.sh "<bench args>" "CFG1" "CFG2" ...  (CFG = "A=1 B=2")
ARGS=$1; shift
for cfg in "$@"; do
  env $cfg python bench.py $ARGS --no-cpu-baseline --no-profile 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', d['ms_per_step'], 'ms', d['value'], 'img/s')"
done
