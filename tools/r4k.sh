cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4k; mkdir -p $O
python -m pytest tests/test_effnet_gpu.py tests/test_effnet_bf16_gpu.py -m gpu -q -x 2>&1 | tail -3
python tools/op_profile.py --streams 1 --steps 4 > $O/op1s.txt 2>/dev/null
head -1 $O/op1s.txt; grep -E "^(k_se_fwd|k_se_bwd|k_se_wgrad)/" $O/op1s.txt | tr '\n' ';'; echo
for i in 1 2; do python bench.py --model Efficient_b0 --precision bf16 --batch 512 --classes 14 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bf16', d['ms_per_step'])"; done
python bench.py --model Efficient_b0 --batch 256 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fp32', d['ms_per_step'])"
