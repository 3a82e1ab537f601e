#!/bin/bash
# round-4 measurement pass behind profiles/r04: for each judged workload the rocprofv3 kernel-trace stats of `bench.py --one-stream`
# (per-kernel averages that must agree with the HIP-event figures of the bench line), the FETCH_SIZE / WRITE_SIZE passes
# (separate rocprofv3 --pmc runs, never combined with other tracing), and the one-stream op profiles of EfficientNet-B0.
# usage: tools/profile_pass.sh <tag>   -> gpurun_out/<tag>_{s1,cf,ebf,ef32}/, gpurun_out/pmc/pmc_traffic.json, gpurun_out/<tag>/op_*.txt
TAG=${1:-r4p}
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
mkdir -p gpurun_out/$TAG gpurun_out/pmc
cp profiles/r04/pmc_traffic.json gpurun_out/pmc/pmc_traffic.json 2>/dev/null
bash tools/prof_stats.sh ${TAG}_s1 --steps 30 --warmup 3 > gpurun_out/$TAG/prof_s1.log 2>&1
bash tools/prof_stats.sh ${TAG}_cf --workload conv_fwd --batch 256 --steps 60 --warmup 3 > gpurun_out/$TAG/prof_cf.log 2>&1
bash tools/prof_stats.sh ${TAG}_ebf --model Efficient_b0 --precision bf16 --batch 512 --classes 14 --steps 20 --warmup 3 > gpurun_out/$TAG/prof_ebf.log 2>&1
bash tools/prof_stats.sh ${TAG}_ef32 --model Efficient_b0 --batch 256 --steps 20 --warmup 3 > gpurun_out/$TAG/prof_ef32.log 2>&1
bash tools/pmc_run.sh s1 "Resnet18/fp32/stage1/bs128/hw224/C5" 3 1 > gpurun_out/$TAG/pmc_s1.log 2>&1
bash tools/pmc_run.sh cf "Resnet18/fp32/conv_fwd/bs256/hw224/C5" 6 1 --workload conv_fwd --batch 256 > gpurun_out/$TAG/pmc_cf.log 2>&1
bash tools/pmc_run.sh ebf "Efficient_b0/bf16/stage1/bs512/hw224/C14" 3 1 --model Efficient_b0 --precision bf16 --batch 512 --classes 14 > gpurun_out/$TAG/pmc_ebf.log 2>&1
bash tools/pmc_run.sh ef32 "Efficient_b0/fp32/stage1/bs256/hw224/C5" 3 1 --model Efficient_b0 --batch 256 > gpurun_out/$TAG/pmc_ef32.log 2>&1
python tools/op_profile.py --precision bf16 --batch 512 --streams 1 > gpurun_out/$TAG/op_profile_one_stream_bf16_bs512.txt 2>/dev/null
python tools/op_profile.py --precision fp32 --batch 256 --streams 1 > gpurun_out/$TAG/op_profile_one_stream_f32_bs256.txt 2>/dev/null
for f in gpurun_out/$TAG/pmc_*.log; do tail -n 2 $f; done
for s in s1 cf ebf ef32; do tail -c 300 gpurun_out/${TAG}_$s/bench.json; echo; done
