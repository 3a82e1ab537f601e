python -m pytest tests/test_effnet_bf16_gpu.py tests/test_golden_r3_gpu.py -m gpu -q -x -k "bf16" 2>&1 | tail -8
bash tools/ab.sh 2 --model Efficient_b0 --precision bf16 --batch 512 --steps 40
