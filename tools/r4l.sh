cd "${GRAFT_REPO_ROOT:-/root/repo}"
export FEDMLP_HIP_LIB=$PWD/fedmlp_amd/libfedmlp_hip_tune.so
b() { python bench.py --no-legs --no-cpu-baseline --sustain-s 0 --steps 40 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'])"; }
b base
FM_TN_FAST=0 b tn_fast0
FM_TN_FAST=1 b tn_fast1
FM_WGRAD_SLOTS=256 b wg256
FM_WGRAD_SLOTS=1024 b wg1024
FM_WGRAD192=0 b wg192_off
FM_SIDE_PRIO=0 b prio0
b base
python -m pytest tests/test_engine_gpu.py -m gpu -q -x -k "three_product" 2>&1 | grep -E "passed|failed|Error|assert" | tail -3
