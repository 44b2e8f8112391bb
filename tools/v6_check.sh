#!/bin/bash
# dev tool: build the engine (optional extra defs in $1), compare the v6 kernel with the shipped 8-wave kernel on the stateful
# path (480p, 21 frames), run the parity tests with VOSPROP_V6=1, and time both.  Everything with timeouts.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/var
hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared -fno-slp-vectorize -L/opt/rocm/lib -lhipblaslt $1 -o gpurun_out/var/chk.so semi-supervised-vos_amd/csrc/engine.hip || exit 1
export VOSPROP_LIB=$PWD/gpurun_out/var/chk.so
echo "== v3 =="; timeout -k 10 120 python tools/prop_bench.py --stateful 2>/dev/null | tail -1 | cut -c1-220
echo "== v6 =="; VOSPROP_V6=1 timeout -k 10 120 python tools/prop_bench.py --stateful 2>/dev/null | tail -1 | cut -c1-220 || exit 1
echo "== v6 vs golden roll-out =="; VOSPROP_V6=1 timeout -k 10 120 python tools/v5_debug.py 2>&1 | tail -6 | cut -c1-200
if [ "${2:-}" = "tests" ]; then
  VOSPROP_V6=1 timeout -k 10 600 python -m pytest tests -m gpu -q -x 2>&1 | tail -4 | cut -c1-200
fi
