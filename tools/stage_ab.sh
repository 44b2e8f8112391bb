#!/bin/bash
# A/B on one box: the shipped dense kernel (only the older wave of every SIMD stages) vs -DVOSPROP_STAGE_ALL (every wave stages).
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared -fno-slp-vectorize -L/opt/rocm/lib -lhipblaslt -DVOSPROP_STAGE_ALL -o gpurun_out/stage_all.so semi-supervised-vos_amd/csrc/engine.hip || exit 1
for i in 1 2 3; do
  echo -n "older-wave staging: "; python tools/prop_bench.py "$@" 2>/dev/null | tail -1 | cut -c1-40
  echo -n "every wave stages:  "; VOSPROP_LIB=$PWD/gpurun_out/stage_all.so python tools/prop_bench.py "$@" 2>/dev/null | tail -1 | cut -c1-40
done
