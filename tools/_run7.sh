mkdir -p gpurun_out
python -m pytest tests -m gpu -q 2>&1 | tail -60 > gpurun_out/pytest_full.log; tail -40 gpurun_out/pytest_full.log
