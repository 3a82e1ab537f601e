# conditioned 64x64 golden replay under kernel variants that only reassociate fp32 sums: prints the max deviations
for cfg in "X=1" "FM_IGEMM_BLOCKS=384" "FM_KS32=0" "FM_TN_FAST=0" "FM_STEM_PACKED=0" "FM_STEM_PACKED=0 FM_IGEMM_BLOCKS=384" "FM_STEM_PACKED=0 FM_KS32=0"; do
  env $cfg python -m pytest tests/test_golden_r2_gpu.py -m gpu -q -k "conditioned_golden_64" > /dev/null 2>&1
  python - "$cfg" <<'PY'
import json,sys
d=json.load(open('gpurun_out/parity_traj_fedmlp64.json'))
print(sys.argv[1], {k:(round(v,5) if isinstance(v,float) else v) for k,v in d['max'].items()}, d["picks_replaced"], d["picks_total"])
PY
done
