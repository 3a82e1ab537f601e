for cfg in "X=1" "FM_BN_BWD_FUSED=0" "FM_SIDE_PRIO=1" "FM_SIDE_PRIO=-1" "FM_BN_BWD_FUSED=0 FM_SIDE_PRIO=1" "FM_BN_BWD_FUSED=0 FM_SIDE_PRIO=-1"; do
  env $cfg python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-profile 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', d['ms_per_step'], 'ms', d['value'], 'img/s')"
done
for cfg in "X=1" "FM_BN_BWD_FUSED=0"; do
  env $cfg python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-profile --one-stream 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('one-stream $cfg', d['ms_per_step'], 'ms', d['value'], 'img/s')"
done
