#!/bin/bash
# copy the judged records of a tools/gpu_r6.sh <tag> pass from gpurun_out/ (scratch) into profiles/r06/
TAG=${1:-r6}; D=profiles/r06
mkdir -p $D
cp gpurun_out/$TAG/bench_stdout.txt $D/bench_gpus1_stdout.txt 2>/dev/null
cp gpurun_out/$TAG/bench_legs.json $D/bench_legs.json 2>/dev/null
cp gpurun_out/$TAG/pytest.log $D/pytest_gpu.log 2>/dev/null
for k in s1:resnet18_stage1_bs128 cf:resnet18_conv_fwd_bs256 ef32:efficient_b0_f32_bs256 ebf:efficient_b0_bf16_bs512; do
  s=${k%%:*}; n=${k##*:}
  [ -f gpurun_out/${TAG}_$s/kernel_stats.csv ] && cp gpurun_out/${TAG}_$s/kernel_stats.csv $D/kernel_stats_one_stream_$n.csv
  [ -f gpurun_out/${TAG}_$s/bench.json ] && tail -1 gpurun_out/${TAG}_$s/bench.json > $D/bench_one_stream_under_rocprof_$n.json
done
cp gpurun_out/$TAG/parity_*.json $D/ 2>/dev/null
cp gpurun_out/$TAG/pmc_traffic.json $D/pmc_traffic.json 2>/dev/null
cp gpurun_out/$TAG/mfma_busy_pconv_conv16_1024imgs.txt $D/ 2>/dev/null
cp gpurun_out/$TAG/pmc_convs_per_layer.txt $D/ 2>/dev/null
ls $D | wc -l
