#!/bin/bash
# L2 locality experiment: kernel time and L2<->fabric traffic of the propagation kernel for VOSPROP_PHASES = 1, 2, 3, 4 (dev tool)
R=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
for P in ${PHASES:-1 2 3 4}; do
  export VOSPROP_PHASES=$P
  t=$(python $R/tools/prop_bench.py --stateful 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f us checksum %.4f' % (d['kernel_us'], d['checksum']))")
  rm -rf $R/gpurun_out/ph_$P
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE FETCH_SIZE --output-format csv -d $R/gpurun_out/ph_$P -- python $R/tools/prop_bench.py --stateful --iters 5 > /dev/null 2>&1
  echo "phases=$P: $t | $(python $R/tools/pmc_summary.py $R/gpurun_out/ph_$P | tr -s ' ' | tr '\n' ';')"
done
