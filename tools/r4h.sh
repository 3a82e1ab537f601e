cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4h; mkdir -p $O
T=$PWD/fedmlp_amd/libfedmlp_hip_tune.so
export FEDMLP_HIP_LIB=$T
for v in 240 96; do
  FM_PW_GEMM_PRO_MINK=$v python tools/op_profile.py --streams 1 --steps 4 > $O/op1s_mink$v.txt 2>/dev/null
  FM_PW_GEMM_PRO_MINK=$v python bench.py --model Efficient_b0 --precision bf16 --batch 512 --classes 14 --steps 30 --warmup 5 --no-cpu-baseline > $O/bench2s_mink$v.json 2>/dev/null
done
for f in $O/op1s_*.txt; do echo "== $f"; head -1 $f; grep -E "^(exp_fwd|proj_fwd|k_se_scale)/" $f | tr '\n' ';'; echo; grep -E "^proj_fwd@(1|2|3|202|203|204) " $f | tr '\n' ';'; echo; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4h/bench*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d['ms_per_step'])
PY
