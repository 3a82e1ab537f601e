#!/bin/bash
# Standard GPU-box pass: parity tests, headline bench, conv-forward leg.  Logs under gpurun_out/<tag>/.
TAG=${1:-check}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q 2>&1 | tail -40 > $OUT/pytest.log
tail -5 $OUT/pytest.log
python bench.py --steps 60 --warmup 5 > $OUT/bench_stage1.json 2> $OUT/bench_stage1.err
tail -c 1500 $OUT/bench_stage1.json
python bench.py --workload conv_fwd --batch 256 --steps 40 --warmup 5 --no-cpu-baseline > $OUT/bench_conv_fwd.json 2> $OUT/bench_conv_fwd.err
tail -c 1200 $OUT/bench_conv_fwd.json
