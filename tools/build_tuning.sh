#!/bin/bash
# Tuning build of the library (-DFM_TUNING: fm_tune() knobs become environment switches) next to the shipped one:
# tune/libfedmlp_hip_tune.so, selected with FEDMLP_HIP_LIB=$PWD/tune/libfedmlp_hip_tune.so (measurements only).  The objects
# live under scratch/ (never pushed to the GPU box); tune/ is deleted at round end (`make clean-scratch`).
cd "$(dirname "$0")/.." || exit 1
B=scratch/build_tune${TUNE_TAG:+_$TUNE_TAG}
OUT=tune/libfedmlp_hip_tune${TUNE_TAG:+_$TUNE_TAG}.so
mkdir -p $B tune
for f in fedmlp_amd/csrc/*.hip; do
  o=$B/$(basename "${f%.hip}").o
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ -n "$(find fedmlp_amd/csrc include -name '*.h' -newer "$o" | head -1)" ]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function -DFM_TUNING $EXTRA_FLAGS -c "$f" -o "$o" &
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT $B/*.o && ls -la $OUT
