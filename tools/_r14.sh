python -m pytest tests/test_golden_r3_gpu.py tests/test_rccl_gpu.py -m gpu -q -k "configs4 or rccl" 2>&1 | tail -5
bash tools/prof_all.sh p3 2>&1 | tail -30
