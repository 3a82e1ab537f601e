cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_engine_gpu.py -m gpu -q -x 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -5
python tools/probe_conv.py 256 1,6,11,16 0,1 2>&1 | grep conv
echo prev; FEDMLP_HIP_LIB=$PWD/fedmlp_amd/libfedmlp_hip_prev.so python tools/probe_conv.py 256 1,6,11,16 0,1 2>&1 | grep conv
b() { python bench.py --no-legs --no-cpu-baseline --sustain-s 0 --steps 60 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'])"; }
for i in 1 2 3; do
b new
FEDMLP_HIP_LIB=$PWD/fedmlp_amd/libfedmlp_hip_prev.so b prev
done
