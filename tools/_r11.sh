for k in 2 4 3; do
  echo "== FM_PW_SMALLK=$k"
  FM_PW_SMALLK=$k python tools/pw_time.py --imgs 1024 --only 2,4,6,8,10 2>/dev/null | grep -E "^ *[0-9]+ " | cut -c1-40,62-76
  FM_PW_SMALLK=$k python bench.py --model Efficient_b0 --precision bf16 --batch 512 --steps 30 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bf16 bs512 two-stream', d['ms_per_step'], 'ms', d['roofline']['frac'])"
done
