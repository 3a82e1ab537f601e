#!/bin/bash
# Driver-level runs kept under profiles/: the configs[4] soak (EfficientNet-B0 bf16, bs 512, C = 14, 200 rounds =
# 100 stage-1 + 100 stage-2, augmentation on) and the 8-client ResNet-18 two-stage run in both stream modes.
# usage: tools/soak.sh <tag>   -> gpurun_out/<tag>/*.log
TAG=${1:-soak}; OUT=gpurun_out/$TAG; mkdir -p $OUT
python -m fedmlp_amd.driver --exp FedMLP --model Efficient_b0 --precision bf16 --batch_size 512 --n_classes 14 \
  --n_clients 1 --n_local 1024 --rounds_warmup 200 --rounds_FedMLP_stage1 100 --augment 1 --pretrained 0 \
  > $OUT/soak_efficient_b0_bf16_bs512_200rounds.log 2>&1
R18="--exp FedMLP --n_clients 8 --n_classes 5 --n_local 1024 --batch_size 128 --rounds_warmup 12 --rounds_FedMLP_stage1 6 --augment 1 --pretrained 0"
python -m fedmlp_amd.driver $R18 > $OUT/driver_resnet18_8clients_bs128_12rounds_augment.log 2>&1
python -m fedmlp_amd.driver $R18 --streams 1 > $OUT/driver_resnet18_8clients_bs128_12rounds_augment_one_stream.log 2>&1
for f in $OUT/*.log; do echo $f; grep round $f | sed -n '1p;3p;$p' | cut -c1-120; done
python - $OUT <<'PY'
import json, sys, glob
a = [json.loads(l) for l in open(sys.argv[1] + "/driver_resnet18_8clients_bs128_12rounds_augment.log") if l.startswith("{")]
b = [json.loads(l) for l in open(sys.argv[1] + "/driver_resnet18_8clients_bs128_12rounds_augment_one_stream.log") if l.startswith("{")]
print("default (two-stream) vs one-stream losses identical:", [x["mean_loss"] for x in a] == [x["mean_loss"] for x in b], len(a), len(b))
PY
