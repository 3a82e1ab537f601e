#!/bin/bash
# Tuning build of the library (-DFM_TUNING: fm_tune() knobs become environment switches) next to the shipped one:
# fedmlp_amd/libfedmlp_hip_tune.so, selected with FEDMLP_HIP_LIB=$PWD/fedmlp_amd/libfedmlp_hip_tune.so (measurements only).
cd "$(dirname "$0")/.." || exit 1
mkdir -p build_tune
for f in fedmlp_amd/csrc/*.hip; do
  o=build_tune/$(basename "${f%.hip}").o
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ -n "$(find fedmlp_amd/csrc include -name '*.h' -newer "$o" | head -1)" ]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function -DFM_TUNING $EXTRA_FLAGS -c "$f" -o "$o" &
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o fedmlp_amd/libfedmlp_hip_tune.so build_tune/*.o && ls -la fedmlp_amd/libfedmlp_hip_tune.so
