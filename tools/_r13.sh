for rep in 1 2; do for v in 1 0 -1; do
  FM_SIDE_PRIO=$v python bench.py --model Efficient_b0 --precision bf16 --batch 512 --steps 40 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('prio $v bf16 bs512', d['ms_per_step'], 'ms')"
done; done
for v in 1 0; do FM_SIDE_PRIO=$v python bench.py --model Efficient_b0 --batch 256 --steps 40 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('prio $v fp32 bs256', d['ms_per_step'], 'ms')"
done
