#!/bin/bash
# Time compile-time variants of the v6 kernel.  Usage: bash tools/v6_variants.sh "<defs1>" "<defs2>" ...
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/var
i=0
for defs in "$@"; do
  i=$((i+1))
  hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared -fno-slp-vectorize -L/opt/rocm/lib -lhipblaslt $defs -o gpurun_out/var/w$i.so semi-supervised-vos_amd/csrc/engine.hip || continue
  echo -n "[$defs] "; VOSPROP_V6=1 VOSPROP_LIB=$PWD/gpurun_out/var/w$i.so timeout -k 10 120 python tools/prop_bench.py --stateful ${BENCH_ARGS:-} 2>/dev/null | tail -1 | cut -c1-60
done
