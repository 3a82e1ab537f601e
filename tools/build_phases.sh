#!/bin/bash
# timing-only build of the library with pconv.hip's phase counters: build/exp/lib_phases.so (read by tools/probe_phases.py)
cd "$(dirname "$0")/.." || exit 1
make -j8 >/dev/null || exit 1
mkdir -p build/exp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function -DPC_PHASES -c fedmlp_amd/csrc/pconv.hip -o build/exp/pconv_phases.o &&
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/exp/lib_phases.so $(ls build/*.o | grep -v '/pconv.o') build/exp/pconv_phases.o && echo build/exp/lib_phases.so
