#!/bin/bash
# build variants and run the small-size stateful-vs-stateless comparison: bash tools/v6_dbg.sh "<defs1>" ...
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/var
i=0
for defs in "$@"; do
  i=$((i+1))
  hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared -fno-slp-vectorize -L/opt/rocm/lib -lhipblaslt $defs -o gpurun_out/var/d$i.so semi-supervised-vos_amd/csrc/engine.hip || continue
  echo "[$defs]"; VOSPROP_V6=1 VOSPROP_LIB=$PWD/gpurun_out/var/d$i.so timeout -k 10 200 python tools/v5_debug2.py 2>&1 | tail -5 | cut -c1-330
done
