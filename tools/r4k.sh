cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4k; mkdir -p $O
python -m pytest tests/test_engine_gpu.py tests/test_effnet_gpu.py tests/test_kat_gpu.py tests/test_tagging_gpu.py -m gpu -q -x 2>&1 | tail -3
python tools/op_profile.py --streams 1 --steps 4 > $O/op1s.txt 2>/dev/null
head -1 $O/op1s.txt; grep -E "^(bn_fwd_finalize|bn_fwd_tensor|bnact_bwd)/" $O/op1s.txt | tr '\n' ';'; echo
for i in 1 2; do python bench.py --model Efficient_b0 --precision bf16 --batch 512 --classes 14 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bf16', d['ms_per_step'])"; done
for i in 1 2; do python bench.py --no-legs --no-cpu-baseline --sustain-s 0 --steps 60 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('r18', d['ms_per_step'])"; done
