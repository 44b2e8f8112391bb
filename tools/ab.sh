#!/bin/bash
# A/B of two builds of libvosprop.so on ONE box (boxes differ by several per cent): the shipped flags against the same sources with
# extra -D switches, alternating runs of tools/prop_bench.py.   Usage (on the GPU box): bash tools/ab.sh "-DVOSPROP_ROWS_EARLY=0" [bench args]
R=$(cd "$(dirname "$0")/.." && pwd)
defs=$1; shift
cd $R/semi-supervised-vos_amd/csrc && hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form=1 $defs -o /tmp/libvos_b.so engine.hip -L/opt/rocm/lib -lhipblaslt -Wl,-rpath,/opt/rocm/lib || exit 1
cd $R
for i in 1 2 3; do
  a=$(python tools/prop_bench.py "$@" | grep -o '"kernel_us": [0-9.]*')
  b=$(VOSPROP_LIB=/tmp/libvos_b.so python tools/prop_bench.py "$@" | grep -o '"kernel_us": [0-9.]*')
  echo "shipped $a | with $defs $b"
done
