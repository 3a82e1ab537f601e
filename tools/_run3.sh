python -m pytest tests/test_engine_gpu.py tests/test_kernels_gpu.py -m gpu -x -q 2>&1 | tail -15 > gpurun_out/t3.log; cat gpurun_out/t3.log
python bench.py --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/b3_default.json 2> gpurun_out/b3_default.err; tail -c 2500 gpurun_out/b3_default.json; tail -3 gpurun_out/b3_default.err
python bench.py --steps 40 --warmup 5 --no-cpu-baseline --one-stream > gpurun_out/b3_one.json 2> gpurun_out/b3_one.err; tail -c 800 gpurun_out/b3_one.json
FM_BN_BWD_FUSED=0 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-profile > gpurun_out/b3_unfused.json 2>/dev/null; tail -c 400 gpurun_out/b3_unfused.json
