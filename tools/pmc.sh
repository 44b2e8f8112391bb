#!/bin/bash
# PMC passes over tools/prop_bench.py (on the GPU box).  Usage: bash tools/pmc.sh <tag> [lib.so] [bench args]
# Counters are collected in separate passes with --kernel-trace only (gpurun refuses pmc + other trace domains).
R=$(cd "$(dirname "$0")/.." && pwd)
tag=$1; lib=${2:-}; shift; shift
cd /tmp && export TMPDIR=/tmp
[ -n "$lib" ] && export VOSPROP_LIB=$lib
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" \
           "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_INST_CYCLES_VMEM SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_INSTS_BRANCH" \
           "GRBM_GUI_ACTIVE FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmc_${tag}_$i -- python $R/tools/prop_bench.py --iters 5 "$@" > $R/gpurun_out/pmc_${tag}_$i.log 2>&1 || echo "pass $i failed"
done
python $R/tools/pmc_summary.py $R/gpurun_out/pmc_${tag}_[0-9] > $R/gpurun_out/pmc_${tag}_summary.txt 2>&1
rm -rf $R/gpurun_out/pmc_${tag}_[0-9]
cat $R/gpurun_out/pmc_${tag}_summary.txt
