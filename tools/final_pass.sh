#!/bin/bash
# Measurement pass behind profiles/r03 for the EfficientNet-B0 configurations (run after tools/gpu_r3.sh <tag>):
# rocprofv3 kernel stats of `bench.py --one-stream`, the FETCH_SIZE / WRITE_SIZE passes, one-stream op profiles.
# usage: tools/final_pass.sh <tag>     -> gpurun_out/<tag>_{ef32,ebf}/, gpurun_out/pmc/pmc_traffic.json, gpurun_out/<tag>/op_*.txt
TAG=${1:-r3}
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
mkdir -p gpurun_out/$TAG gpurun_out/pmc
cp profiles/r03/pmc_traffic.json gpurun_out/pmc/pmc_traffic.json 2>/dev/null      # the other workloads' entries are kept
bash tools/prof_stats.sh ${TAG}_ebf --model Efficient_b0 --precision bf16 --batch 512 --steps 20 --warmup 3 > gpurun_out/$TAG/prof_ebf.log 2>&1
bash tools/prof_stats.sh ${TAG}_ef32 --model Efficient_b0 --batch 256 --steps 20 --warmup 3 > gpurun_out/$TAG/prof_ef32.log 2>&1
bash tools/pmc_run.sh ebf "Efficient_b0/bf16/stage1/bs512/hw224/C5" 3 1 --model Efficient_b0 --precision bf16 --batch 512 > gpurun_out/$TAG/pmc_ebf.log 2>&1
bash tools/pmc_run.sh ef32 "Efficient_b0/fp32/stage1/bs256/hw224/C5" 3 1 --model Efficient_b0 --batch 256 > gpurun_out/$TAG/pmc_ef32.log 2>&1
python tools/op_profile.py --precision bf16 --batch 512 --streams 1 > gpurun_out/$TAG/op_profile_one_stream_bf16_bs512.txt 2>/dev/null
python tools/op_profile.py --precision fp32 --batch 256 --streams 1 > gpurun_out/$TAG/op_profile_one_stream_f32_bs256.txt 2>/dev/null
for f in gpurun_out/$TAG/pmc_ebf.log gpurun_out/$TAG/pmc_ef32.log; do tail -n 3 $f; done
