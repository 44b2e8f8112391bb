#!/bin/bash
# builds of the wide kernel with different -D switches against each other on one box (VOSPROP_WIDE=1 for all)
R=$(cd "$(dirname "$0")/.." && pwd)
cd $R/semi-supervised-vos_amd/csrc
i=0
for defs in "$@"; do
  i=$((i+1))
  hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form=1 $defs -o /tmp/libvos_w$i.so engine.hip -L/opt/rocm/lib -lhipblaslt -Wl,-rpath,/opt/rocm/lib || exit 1
done
cd $R
for rep in 1 2 3; do
  line="8-wave $(python tools/prop_bench.py --stateful | grep -o '"kernel_us": [0-9.]*' | cut -c14-19)"
  i=0
  for defs in "$@"; do
    i=$((i+1))
    line="$line | [$defs] $(VOSPROP_WIDE=1 VOSPROP_LIB=/tmp/libvos_w$i.so python tools/prop_bench.py --stateful | grep -o '"kernel_us": [0-9.]*' | cut -c14-19)"
  done
  echo "$line"
done
