#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
mkdir -p gpurun_out
timeout 300 python3 -m pytest tests/test_kat_gpu.py -x -q 2>&1 | tail -4 > gpurun_out/quick2.txt
timeout 900 python3 -m pytest tests/test_golden_r4_gpu.py -x -q -k "shared_relu_masks" 2>&1 | tail -6 >> gpurun_out/quick2.txt
python3 -c "
import json; d=json.load(open('gpurun_out/parity_step_full_shared_masks.json')); print({k:d[k] for k in ('mask_flips','maxpool_choices_differing','stem_tensors','worst_below_the_stem','loss_rel_err')}); print(d.get('float64'))" >> gpurun_out/quick2.txt 2>&1
timeout 200 python3 bench.py --legs cos_tag_c5,cos_tag_c14 --steps 10 --warmup 3 --sustain-s 0 --no-cpu-baseline 2>&1 | grep '"leg"' | cut -c1-300 >> gpurun_out/quick2.txt
cat gpurun_out/quick2.txt
