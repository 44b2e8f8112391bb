#!/bin/bash
# Time every build/variants/libvos_*.so (tools/mask_variants.sh) with tools/prop_bench.py --stateful on ONE box, two interleaved
# rounds.  Usage (GPU box): bash tools/mask_ab.sh [bench args]
R=$(cd "$(dirname "$0")/.." && pwd)
cd $R
for round in 1 2; do
  for lib in build/variants/libvos_*.so; do
    n=$(basename $lib .so); n=${n#libvos_}
    us=$(VOSPROP_LIB=$R/$lib timeout -k 5 120 python tools/prop_bench.py --stateful "$@" 2>/dev/null | grep -o '"kernel_us": [0-9.]*' | cut -c14-20)
    echo "$n $us"
  done
done | sort | awk '{a[$1]=a[$1]" "$2} END {for (k in a) print k":"a[k]}' | sort
