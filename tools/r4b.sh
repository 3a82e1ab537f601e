cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r4b
python -m pytest tests/test_effnet_bf16_gpu.py -m gpu -q -x 2>&1 | tail -15 > gpurun_out/r4b/pytest_bf16.log; tail -4 gpurun_out/r4b/pytest_bf16.log
T=$PWD/fedmlp_amd/libfedmlp_hip_tune.so
FEDMLP_HIP_LIB=$T FM_PW_GEMM=0 python tools/pw_time.py > gpurun_out/r4b/pw_old.txt 2>&1
FEDMLP_HIP_LIB=$T FM_PW_GEMM=1 python tools/pw_time.py > gpurun_out/r4b/pw_gemm_ns2.txt 2>&1
FEDMLP_HIP_LIB=$T FM_PW_GEMM=1 FM_PW_GEMM_NS=3 python tools/pw_time.py > gpurun_out/r4b/pw_gemm_ns3.txt 2>&1
python bench.py --model Efficient_b0 --precision bf16 --batch 512 --classes 14 --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r4b/bench_ebf.json 2> gpurun_out/r4b/bench_ebf.err
FEDMLP_HIP_LIB=$T FM_PW_GEMM=0 python bench.py --model Efficient_b0 --precision bf16 --batch 512 --classes 14 --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r4b/bench_ebf_old.json 2>/dev/null
python -c "
import json
for f in ('bench_ebf','bench_ebf_old'):
    d=json.loads(open('gpurun_out/r4b/%s.json'%f).read().strip().splitlines()[-1]); print(f, d['ms_per_step'], d['roofline']['frac'])
"
