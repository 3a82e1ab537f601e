cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -8
python tools/probe_conv.py 256 6,11,16 0,1 2>&1 | grep conv
for i in 1 2; do python bench.py --no-legs --no-cpu-baseline --sustain-s 0 --steps 40 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('r18', d['ms_per_step'])"; done
FM_MFMA_SPLIT=9 python bench.py --no-legs --no-cpu-baseline --sustain-s 0 --steps 40 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('r18 x9', d['ms_per_step'])"
