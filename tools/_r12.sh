for rep in 1 2; do for v in head mode; do
  cp build/ab/lib_$v.so fedmlp_amd/libfedmlp_hip.so
  python bench.py --model Efficient_b0 --precision bf16 --batch 512 --steps 40 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v bf16 bs512', d['ms_per_step'], 'ms', d['roofline']['frac'])"
done; done
