cd "${GRAFT_REPO_ROOT:-/root/repo}"
b() { python bench.py --no-legs --no-cpu-baseline --sustain-s 0 --steps 60 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'])"; }
for i in 1 2 3; do
b new
FEDMLP_HIP_LIB=$PWD/fedmlp_amd/libfedmlp_hip_prev.so b prev
done
