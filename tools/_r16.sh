python -m pytest tests/test_effnet_gpu.py tests/test_golden_r3_gpu.py -m gpu -q -x -k "not bf16" 2>&1 | tail -5
bash tools/ab.sh 2 --model Efficient_b0 --batch 256 --steps 40
