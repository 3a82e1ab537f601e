cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4g; mkdir -p $O
python -m pytest tests/test_effnet_bf16_gpu.py -m gpu -q -x 2>&1 | tail -8 > $O/pytest_bf16.log; tail -4 $O/pytest_bf16.log
T=$PWD/fedmlp_amd/libfedmlp_hip_tune.so
export FEDMLP_HIP_LIB=$T
for v in 0 1; do
  FM_PW_GEMM_PRO=$v python tools/op_profile.py --streams 1 --steps 4 > $O/op1s_pro$v.txt 2>/dev/null
  FM_PW_GEMM_PRO=$v python bench.py --model Efficient_b0 --precision bf16 --batch 512 --classes 14 --steps 30 --warmup 5 --no-cpu-baseline > $O/bench2s_pro$v.json 2>/dev/null
done
for f in $O/op1s_*.txt; do echo "== $f"; head -1 $f; grep -E "^(exp_fwd|proj_fwd|k_se_scale)/" $f | tr '\n' ';'; echo; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4g/bench*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d['ms_per_step'])
PY
