#!/bin/bash
# L2<->fabric traffic of the propagation kernel for two settings of VOSPROP_PHASES (dev tool; stateless bench like tools/pmc.sh)
R=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
for P in ${PHASES:-1 2}; do
  export VOSPROP_PHASES=$P
  rm -rf $R/gpurun_out/ph_$P
  timeout -k 10 120 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE FETCH_SIZE --output-format csv -d $R/gpurun_out/ph_$P -- python $R/tools/prop_bench.py --iters 5 > $R/gpurun_out/ph_$P.log 2>&1 || echo "pass failed (see gpurun_out/ph_$P.log)"
  echo "phases=$P: $(python $R/tools/pmc_summary.py $R/gpurun_out/ph_$P | tr -s ' ' | tr '\n' ';')"
done
