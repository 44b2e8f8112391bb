#!/bin/bash
# A/B of the work maps on the same box: bench.py value + kernel time (dev tool)
cd "$(dirname "$0")/.."
run() { python bench.py --no-cpu-baseline --steps 160 --warmup 16 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['value'],1), 'fps  kernel', round(d['roofline']['kernel_us'],1), 'us')"; }
VOSPROP_MAP=streamk run streamk
VOSPROP_SEGCOST=3 run lockstep_c3
VOSPROP_SEGCOST=6 run lockstep_c6
VOSPROP_SEGCOST=10 run lockstep_c10
VOSPROP_MAP=streamk run streamk
VOSPROP_SEGCOST=6 run lockstep_c6
