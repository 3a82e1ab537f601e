cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4i; mkdir -p $O
python -m pytest tests/test_effnet_bf16_gpu.py tests/test_golden_r3_gpu.py -m gpu -q -k "bf16" 2>&1 | tail -4
T=$PWD/fedmlp_amd/libfedmlp_hip_tune.so
export FEDMLP_HIP_LIB=$T
for v in 128 256; do
  FM_PW_GEMM_PRO_MAXM=$v python tools/op_profile.py --streams 1 --steps 4 > $O/op1s_maxm$v.txt 2>/dev/null
  FM_PW_GEMM_PRO_MAXM=$v python bench.py --model Efficient_b0 --precision bf16 --batch 512 --classes 14 --steps 30 --warmup 5 --no-cpu-baseline > $O/bench2s_maxm$v.json 2>/dev/null
done
for f in $O/op1s_*.txt; do echo "== $f"; head -1 $f; grep -E "^(proj_fwd|k_se_scale|proj_wgrad|proj_dgrad|k_se_bwd|bnact_bwd)/" $f | tr '\n' ';'; echo; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4i/bench*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d['ms_per_step'])
PY
