#!/bin/bash
# Build variants of libvosprop.so whose prop_mask_kernel loop comes from another run of the generator (ablations, schedule
# options) into build/variants/ - locally, so that they travel to the GPU box with the snapshot.
#   bash tools/mask_variants.sh name1:"--ablate no_dma" name2:"--opt dma_gaps={3:[2,6,10],2:[3,9]}" ...
R=$(cd "$(dirname "$0")/.." && pwd)
GEN=${GEN:-tools/gen_mask_loop.py}; INCMACRO=${INCMACRO:-VOSPROP_MASK_LOOP_INC}
mkdir -p $R/build/variants
pids=()
for spec in "$@"; do
  name=${spec%%:*}; args=${spec#*:}
  (
    inc=$R/build/variants/loop_$name.inc
    python $R/$GEN --out $inc $args > /dev/null || exit 1
    cd $R/semi-supervised-vos_amd/csrc && /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared -fno-slp-vectorize \
      -mllvm -amdgpu-mfma-vgpr-form=1 -D$INCMACRO="\"$inc\"" -o $R/build/variants/libvos_$name.so engine.hip \
      -L/opt/rocm/lib -lhipblaslt -Wl,-rpath,/opt/rocm/lib && echo "built $name"
  ) &
  pids+=($!)
  if [ ${#pids[@]} -ge 4 ]; then wait ${pids[0]}; pids=("${pids[@]:1}"); fi
done
wait
