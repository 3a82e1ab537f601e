cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4d; mkdir -p $O
python -m pytest tests/test_effnet_bf16_gpu.py -m gpu -q -x -k "pw_conv or prologue or forward_eval" 2>&1 | tail -5 > $O/pytest_bf16.log; tail -3 $O/pytest_bf16.log
T=$PWD/fedmlp_amd/libfedmlp_hip_tune.so
export FEDMLP_HIP_LIB=$T
FM_PW_FULLM=0 python tools/pw_time.py --only 2,4,6,8,10 > $O/pw_full0.txt 2>&1
FM_PW_FULLM=1 python tools/pw_time.py --only 2,4,6,8,10 > $O/pw_full1.txt 2>&1
FM_PW_FULLM=1 FM_PW_FULLM_P6=4 FM_PW_FULLM_144=5 python tools/pw_time.py --only 2,4,6,8,10 > $O/pw_full1_b.txt 2>&1
FM_PW_FULLM=1 FM_PW_FULLM_P9=4 python tools/pw_time.py --only 4 > $O/pw_full1_c.txt 2>&1
for v in "0 0 9 2" "0 1 9 2" "0 1 5 4" "1 1 5 4"; do set -- $v
  FM_PW_GEMM=$1 FM_PW_FULLM=$2 FM_PW_FULLM_144=$3 FM_PW_FULLM_P6=$4 python bench.py --model Efficient_b0 --precision bf16 --batch 512 --classes 14 --steps 20 --warmup 4 --no-cpu-baseline --one-stream > $O/bench1s_g$1_f$2_$3_$4.json 2>/dev/null
  FM_PW_GEMM=$1 FM_PW_FULLM=$2 FM_PW_FULLM_144=$3 FM_PW_FULLM_P6=$4 python bench.py --model Efficient_b0 --precision bf16 --batch 512 --classes 14 --steps 30 --warmup 5 --no-cpu-baseline > $O/bench2s_g$1_f$2_$3_$4.json 2>/dev/null
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4d/bench*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d['ms_per_step'])
    except Exception as e: print(f, 'FAILED', e)
PY
tail -n 7 $O/pw_full0.txt $O/pw_full1.txt $O/pw_full1_b.txt $O/pw_full1_c.txt
