#!/bin/bash
# stamp build of v6 with extra defs: bash tools/v6_stamps.sh "<defs1>" "<defs2>" ...
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/var
i=0
for defs in "$@"; do
  i=$((i+1))
  hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared -fno-slp-vectorize -L/opt/rocm/lib -lhipblaslt -DVOSPROP_STAMP $defs -o gpurun_out/var/s$i.so semi-supervised-vos_amd/csrc/engine.hip || continue
  echo "[$defs] $(VOSPROP_V6=1 VOSPROP_LIB=$PWD/gpurun_out/var/s$i.so timeout -k 10 120 python tools/prop_bench.py --stateful 2>/dev/null | grep -E 'kernel_us|group A' | sed -e 's/, "tflops.*//' -e 's/group A (waves 0-3)://' | tr '\n' ' ')"
done
