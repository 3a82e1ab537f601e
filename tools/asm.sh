#!/bin/bash
# device ISA of one source: tools/asm.sh <name> [extra flags]  -> scratch/asm/<name>.s
cd "$(dirname "$0")/.." || exit 1
n=$1; shift
mkdir -p scratch/asm
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-function --cuda-device-only -S "$@" fedmlp_amd/csrc/$n.hip -o scratch/asm/$n.s
