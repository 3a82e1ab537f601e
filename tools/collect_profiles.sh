#!/bin/bash
# copy the judged records of a tools/gpu_full.sh + tools/prof_stats.sh pass from gpurun_out/ (scratch) into profiles/r03/
# usage: tools/collect_profiles.sh <tag>      (after tools/gpu_r3.sh <tag> and tools/final_pass.sh <tag>: gpurun_out/<tag>/, gpurun_out/<tag>_{s1,cf,ef32,ebf}/)
TAG=$1; D=profiles/r03
mkdir -p $D
cp gpurun_out/$TAG/bench_*.json $D/ 2>/dev/null
cp gpurun_out/$TAG/pytest.log $D/pytest_gpu.log 2>/dev/null
for k in s1:resnet18_stage1_bs128 cf:resnet18_conv_fwd_bs256 ef32:efficient_b0_f32_bs256 ebf:efficient_b0_bf16_bs512; do
  s=${k%%:*}; n=${k##*:}
  [ -f gpurun_out/${TAG}_$s/kernel_stats.csv ] && cp gpurun_out/${TAG}_$s/kernel_stats.csv $D/kernel_stats_one_stream_$n.csv
  [ -f gpurun_out/${TAG}_$s/bench.json ] && cp gpurun_out/${TAG}_$s/bench.json $D/bench_one_stream_under_rocprof_$n.json
done
cp gpurun_out/parity_*.json $D/ 2>/dev/null
cp gpurun_out/$TAG/op_profile_one_stream_*.txt $D/ 2>/dev/null
[ -f gpurun_out/pmc/pmc_traffic.json ] && cp gpurun_out/pmc/pmc_traffic.json $D/pmc_traffic.json
ls $D
