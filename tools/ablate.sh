#!/bin/bash
# Build timing-only variants of the propagation kernel (results are WRONG by construction) and time them.
# Usage (on the GPU box): bash tools/ablate.sh "0 1 2 4 12 13 14 15"
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/ablate
for m in ${1:-0 1 2 4 12}; do
  hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared -fno-slp-vectorize -L/opt/rocm/lib -lhipblaslt -DVOSPROP_ABLATE=$m -o gpurun_out/ablate/libvosprop_ab$m.so semi-supervised-vos_amd/csrc/engine.hip
  echo -n "ablate=$m "; VOSPROP_LIB=$PWD/gpurun_out/ablate/libvosprop_ab$m.so python tools/prop_bench.py ${2:-} 2>/dev/null | tail -1
done
