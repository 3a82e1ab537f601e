#!/bin/bash
# timing-only build of the library with pconv.hip's phase counters (and the tuning knobs): tune/lib_phases.so (tools/probe_phases.py)
cd "$(dirname "$0")/.." || exit 1
[ -f tune/libfedmlp_hip_tune.so ] || tools/build_tuning.sh >/dev/null || exit 1
mkdir -p scratch/phases tune
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function -DFM_TUNING -DPC_PHASES -c fedmlp_amd/csrc/pconv.hip -o scratch/phases/pconv_phases.o &&
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tune/lib_phases.so $(ls scratch/build_tune/*.o | grep -v '/pconv.o') scratch/phases/pconv_phases.o && echo tune/lib_phases.so
