cd "${GRAFT_REPO_ROOT:-/root/repo}"
bash tools/prof_stats.sh r4s6_s1 --steps 30 --warmup 3 --no-legs --sustain-s 0 > gpurun_out/r4s6_s1.log 2>&1
head -24 gpurun_out/r4s6_s1/kernel_stats.csv | cut -c1-150
timeout 600 python -m pytest tests/test_eval_gpu.py -m gpu -q -x -k "converged" 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
cat gpurun_out/parity_map_converged.json; echo
python bench.py --legs stage1_fp32_mfma_pipe,conv_fwd_bs256 --sustain-s 3 > gpurun_out/r4s6_bench.json 2> gpurun_out/r4s6_bench.err; tail -3 gpurun_out/r4s6_bench.err; cut -c1-1500 gpurun_out/r4s6_bench.json
