python -m pytest tests/test_golden_r3_gpu.py tests/test_golden_r2_gpu.py -m gpu -q --tb=short 2>&1 | tail -80 > gpurun_out/pytest_r3.log; tail -70 gpurun_out/pytest_r3.log
