#!/bin/bash
# Time compile-time variants of the propagation kernel.  Usage: bash tools/variants.sh "<defs1>" "<defs2>" ...
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/var
i=0
for defs in "$@"; do
  i=$((i+1))
  hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared -fno-slp-vectorize -L/opt/rocm/lib -lhipblaslt $defs -o gpurun_out/var/v$i.so semi-supervised-vos_amd/csrc/engine.hip || continue
  echo -n "[$defs] "; VOSPROP_LIB=$PWD/gpurun_out/var/v$i.so python tools/prop_bench.py --stateful ${BENCH_ARGS:-} 2>/dev/null | tail -1 | cut -c1-100
done
