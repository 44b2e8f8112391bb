#!/bin/bash
# build a variant of the engine and run the stateful-path parity tests against it (dev tool)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/var
hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared -fno-slp-vectorize -L/opt/rocm/lib -lhipblaslt $1 -o gpurun_out/var/chk.so semi-supervised-vos_amd/csrc/engine.hip || exit 1
VOSPROP_LIB=$PWD/gpurun_out/var/chk.so timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "rollout_vs_golden or smoke" 2>&1 | tail -4 | cut -c1-150
