#!/bin/bash
# round-3 GPU pass: full test suite + the benches of every BASELINE config (default two-stream mode) -> gpurun_out/<tag>/
TAG=${1:-r3}; OUT=gpurun_out/$TAG; mkdir -p $OUT
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
python -m pytest tests -m gpu -q 2>&1 | tail -40 > $OUT/pytest.log; tail -5 $OUT/pytest.log
python bench.py > $OUT/bench_resnet18_stage1_bs128.json 2> $OUT/err1.log
python bench.py --workload conv_fwd --batch 256 --steps 60 --no-cpu-baseline > $OUT/bench_resnet18_conv_fwd_bs256.json 2> $OUT/err2.log
python bench.py --model Efficient_b0 --batch 256 --steps 40 --no-cpu-baseline > $OUT/bench_efficient_b0_f32_bs256.json 2> $OUT/err3.log
python bench.py --model Efficient_b0 --precision bf16 --batch 512 --steps 40 --no-cpu-baseline > $OUT/bench_efficient_b0_bf16_bs512.json 2> $OUT/err4.log
python bench.py --classes 14 --steps 40 --no-cpu-baseline > $OUT/bench_resnet18_stage1_c14.json 2> $OUT/err5.log
python bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu-baseline --no-profile > $OUT/bench_gpus1.json 2> $OUT/err6.log
for f in $OUT/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r=d.get("roofline") or {}
    print(sys.argv[1].split('/')[-1], d["value"], "img/s", d["ms_per_step"], "ms", r.get("bound"), r.get("frac"), r.get("whole_step_frac"))
except Exception as ex:
    print(sys.argv[1], "FAILED", ex)
PY
done
