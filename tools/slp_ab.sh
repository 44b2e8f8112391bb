#!/bin/bash
# A/B: the shipped flags vs a build WITHOUT -fno-slp-vectorize (packed f32 VALU instructions in the softmax rows), same box.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared -L/opt/rocm/lib -lhipblaslt -o gpurun_out/slp.so semi-supervised-vos_amd/csrc/engine.hip || exit 1
echo "shipped:"; python tools/prop_bench.py "$@" 2>/dev/null | tail -1
echo "slp-vectorize on:"; VOSPROP_LIB=$PWD/gpurun_out/slp.so python tools/prop_bench.py "$@" 2>/dev/null | tail -1
