bash tools/gpu_r3.sh r3f 2>&1 | tail -12
bash tools/prof_all.sh p3f 2>&1 | tail -25
