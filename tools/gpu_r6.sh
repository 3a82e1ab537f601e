#!/bin/bash
# round-6 GPU pass: full test suite, the bench line with legs, one-stream rocprofv3 kernel statistics, PMC traffic passes and the
# matrix-pipe counters.  Every command has its own timeout (a hung kernel must not eat the GPU budget).
# usage: tools/gpu_r6.sh <tag> [skip-tests]   -> gpurun_out/<tag>/...
TAG=${1:-r6}; SKIP=$2
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/$TAG; mkdir -p $OUT gpurun_out/pmc
if [ -z "$SKIP" ]; then
  timeout 2400 python3 -m pytest tests -m gpu -q 2>&1 | tail -40 > $OUT/pytest.log; tail -4 $OUT/pytest.log
  cp gpurun_out/parity_*.json $OUT/ 2>/dev/null
fi
timeout 600 python3 bench.py > $OUT/bench_stdout.txt 2> $OUT/bench_err.log; tail -1 $OUT/bench_stdout.txt | cut -c1-600
cp bench_legs.json $OUT/bench_legs.json 2>/dev/null
timeout 240 bash tools/prof_stats.sh ${TAG}_s1 --steps 30 --warmup 3 --no-legs --sustain-s 0 > $OUT/prof_s1.log 2>&1
timeout 240 bash tools/prof_stats.sh ${TAG}_cf --workload conv_fwd --batch 256 --steps 60 --warmup 3 > $OUT/prof_cf.log 2>&1
timeout 300 bash tools/prof_stats.sh ${TAG}_ebf --model Efficient_b0 --precision bf16 --batch 512 --classes 14 --steps 20 --warmup 3 > $OUT/prof_ebf.log 2>&1
timeout 300 bash tools/prof_stats.sh ${TAG}_ef32 --model Efficient_b0 --batch 256 --steps 20 --warmup 3 > $OUT/prof_ef32.log 2>&1
cp profiles/r05/pmc_traffic.json gpurun_out/pmc/pmc_traffic.json 2>/dev/null
timeout 400 bash tools/pmc_run.sh s1 "Resnet18/fp32/stage1/bs128/hw224/C5" 3 1 > $OUT/pmc_s1.log 2>&1
timeout 300 bash tools/pmc_run.sh cf "Resnet18/fp32/conv_fwd/bs256/hw224/C5" 6 1 --workload conv_fwd --batch 256 > $OUT/pmc_cf.log 2>&1
timeout 400 bash tools/pmc_run.sh ebf "Efficient_b0/bf16/stage1/bs512/hw224/C14" 3 1 --model Efficient_b0 --precision bf16 --batch 512 --classes 14 > $OUT/pmc_ebf.log 2>&1
timeout 400 bash tools/pmc_run.sh ef32 "Efficient_b0/fp32/stage1/bs256/hw224/C5" 3 1 --model Efficient_b0 --batch 256 > $OUT/pmc_ef32.log 2>&1
cp gpurun_out/pmc/pmc_traffic.json $OUT/pmc_traffic.json 2>/dev/null
FM_DEBUG_REUSE_PLANES=1 timeout 300 bash tools/mfma_busy.sh 16 > $OUT/mfma_busy.log 2>&1; cp gpurun_out/mfma_busy.txt $OUT/mfma_busy_pconv_conv16_1024imgs.txt 2>/dev/null
timeout 300 bash tools/pmc_convs.sh $TAG 256 1,6,11,16 0,1 > $OUT/pmc_convs.log 2>&1; cp gpurun_out/pmc_convs_$TAG.txt $OUT/pmc_convs_per_layer.txt 2>/dev/null
for f in $OUT/pmc_*.log; do tail -n 2 $f; done
for s in s1 cf ebf ef32; do tail -c 250 gpurun_out/${TAG}_$s/bench.json 2>/dev/null; echo; done
cat $OUT/mfma_busy_pconv_conv16_1024imgs.txt 2>/dev/null
