#!/bin/bash
# Timing-only variants of the in-wave pipelined dense kernel (csrc/prop_dense.h; results are WRONG by construction).
# Usage (on the GPU box): bash tools/dense_ablate.sh "0 1 2 4 8 16" [prop_bench args]
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/dablate
for m in ${1:-0 1 2 4 8 16}; do
  hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form=1 -L/opt/rocm/lib -lhipblaslt -DVOSPROP_DABLATE=$m -o gpurun_out/dablate/lib$m.so semi-supervised-vos_amd/csrc/engine.hip
  echo -n "dense ablate=$m "; VOSPROP_LIB=$PWD/gpurun_out/dablate/lib$m.so python tools/prop_bench.py --stateful ${2:-} 2>/dev/null | tail -1 | cut -c1-110
done
