cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_engine_gpu.py -m gpu -q -x 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -8
export FEDMLP_HIP_LIB=$PWD/fedmlp_amd/libfedmlp_hip_tune.so
for w in 1 0; do
echo "== FM_WPLANES=$w"
FM_WPLANES=$w python tools/probe_conv.py 256 6,11,16 0,1 2>&1 | grep conv
FM_WPLANES=$w python bench.py --no-legs --no-cpu-baseline --sustain-s 0 --steps 40 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('r18 planes=$w', d['ms_per_step'])"
done
FM_WPLANES=1 FM_MFMA_SPLIT=9 python bench.py --no-legs --no-cpu-baseline --sustain-s 0 --steps 40 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('r18 planes=1 x9', d['ms_per_step'])"
