#!/bin/bash
# Diagnostic build with in-kernel s_memtime stamps (never timed, never shipped): cycle shares of the loop phases.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form=1 -L/opt/rocm/lib -lhipblaslt -DVOSPROP_STAMP ${STAMP_DEFS:-} -o gpurun_out/stamp.so semi-supervised-vos_amd/csrc/engine.hip || exit 1
VOSPROP_LIB=$PWD/gpurun_out/stamp.so python tools/prop_bench.py "$@" 2>/dev/null | tail -9
