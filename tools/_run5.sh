python tools/op_profile.py --precision bf16 --batch 512 --streams 1 > gpurun_out/op5_bf16.txt 2>&1; head -24 gpurun_out/op5_bf16.txt
python bench.py --model Efficient_b0 --precision bf16 --batch 512 --steps 30 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bf16 bs512 two-stream', d['ms_per_step'], 'ms', d['roofline']['frac'])"
python bench.py --model Efficient_b0 --batch 256 --steps 30 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fp32 bs256 two-stream', d['ms_per_step'], 'ms', d['roofline']['frac'])"
python -m pytest tests/test_effnet_bf16_gpu.py tests/test_effnet_gpu.py -m gpu -x -q 2>&1 | tail -5
