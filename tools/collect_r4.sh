#!/bin/bash
# copy the judged records of a tools/gpu_r4.sh <tag> + tools/profile_pass.sh <ptag> pass from gpurun_out/ (scratch) into profiles/r04/
# usage: tools/collect_r4.sh <tag> <ptag>
TAG=$1; PT=$2; D=profiles/r04
mkdir -p $D
cp gpurun_out/$TAG/bench.json $D/bench_gpus1_with_legs.json 2>/dev/null
cp gpurun_out/$TAG/pytest.log $D/pytest_gpu.log 2>/dev/null
for k in s1:resnet18_stage1_bs128 cf:resnet18_conv_fwd_bs256 ef32:efficient_b0_f32_bs256 ebf:efficient_b0_bf16_bs512; do
  s=${k%%:*}; n=${k##*:}
  [ -f gpurun_out/${PT}_$s/kernel_stats.csv ] && cp gpurun_out/${PT}_$s/kernel_stats.csv $D/kernel_stats_one_stream_$n.csv
  [ -f gpurun_out/${PT}_$s/bench.json ] && cp gpurun_out/${PT}_$s/bench.json $D/bench_one_stream_under_rocprof_$n.json
done
cp gpurun_out/parity_*.json $D/ 2>/dev/null
cp gpurun_out/$PT/op_profile_one_stream_*.txt $D/ 2>/dev/null
[ -f gpurun_out/pmc/pmc_traffic.json ] && cp gpurun_out/pmc/pmc_traffic.json $D/pmc_traffic.json
ls $D | wc -l
