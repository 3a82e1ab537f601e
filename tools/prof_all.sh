#!/bin/bash
# rocprofv3 stats + PMC traffic of the four single-GPU workloads, op profiles.  usage: tools/prof_all.sh <tag>
# (ONE gpurun call: gpurun_out/ starts empty on every box, so pmc_traffic.json only holds all four keys this way)
TAG=$1
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /tmp/envprobe -- python3 -c "import os; print('profiler env:', sorted(k for k in os.environ if 'ROCP' in k.upper() or k == 'LD_PRELOAD'))" 2>/dev/null | grep "profiler env"
bash tools/prof_stats.sh ${TAG}_s1 > /dev/null 2>&1
bash tools/prof_stats.sh ${TAG}_cf --workload conv_fwd --batch 256 --steps 60 > /dev/null 2>&1
bash tools/prof_stats.sh ${TAG}_ef32 --model Efficient_b0 --batch 256 --steps 40 > /dev/null 2>&1
bash tools/prof_stats.sh ${TAG}_ebf --model Efficient_b0 --precision bf16 --batch 512 --steps 40 > /dev/null 2>&1
for t in s1 cf ef32 ebf; do head -3 gpurun_out/${TAG}_$t/kernel_stats.csv | cut -c1-150; tail -c 250 gpurun_out/${TAG}_$t/bench.json; echo; done
bash tools/pmc_run.sh s1 Resnet18/fp32/stage1/bs128/hw224/C5 6 2 2>&1 | tail -3
bash tools/pmc_run.sh cf Resnet18/fp32/conv_fwd/bs256/hw224/C5 6 2 --workload conv_fwd --batch 256 2>&1 | tail -3
bash tools/pmc_run.sh ef32 Efficient_b0/fp32/stage1/bs256/hw224/C5 4 2 --model Efficient_b0 --batch 256 2>&1 | tail -3
bash tools/pmc_run.sh ebf Efficient_b0/bf16/stage1/bs512/hw224/C5 4 2 --model Efficient_b0 --precision bf16 --batch 512 2>&1 | tail -3
python tools/op_profile.py --precision bf16 --batch 512 --streams 1 > gpurun_out/${TAG}_op_bf16.txt 2>&1
python tools/op_profile.py --precision fp32 --batch 256 --streams 1 > gpurun_out/${TAG}_op_f32.txt 2>&1
head -12 gpurun_out/${TAG}_op_bf16.txt
