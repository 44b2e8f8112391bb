#!/bin/bash
# Round-2 evidence in one go (on the GPU box); outputs under gpurun_out/r02/ (copy what is cited into profiles/).
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out/r02
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
PARTS=" ${R02_PARTS:-1 2 3 4} "   # R02_PARTS="3 4": only the traffic and SQ-counter passes
# 1. kernel stats of the default bench command
if [[ "$PARTS" == *" 1 "* ]]; then
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_prof -- python $R/bench.py > $O/bench_under_rocprofv3.json 2> $O/bench_prof.err || echo "bench profile failed"
cp $(find $O/bench_prof -name '*kernel_stats.csv' | head -1) $O/bench_kernel_stats.csv 2>/dev/null
rm -rf $O/bench_prof
echo "bench profile done"
fi
# 2. clock / cache state of the propagation launches by position after an encoder batch
if [[ "$PARTS" == *" 2 "* ]]; then
for set in "GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/ramp_$tag -- python $R/bench.py --no-cpu-baseline --no-end-to-end --steps 256 --warmup 16 > $O/ramp_$tag.json 2> $O/ramp_$tag.err || echo "ramp $tag failed"
  python $R/tools/clock_ramp.py $O/ramp_$tag 36 256 > $O/clock_ramp_$tag.txt 2>&1
  rm -rf $O/ramp_$tag
done
echo "ramp done"
fi
# 3. traffic of the three propagation workloads
cd $R
if [[ "$PARTS" == *" 3 "* ]]; then
bash tools/traffic_pmc.sh davis480p_r50_dense --stateful > $O/traffic_480p.log 2>&1
bash tools/traffic_pmc.sh ytvos720p_r50_dense --stateful --hd 90 --wd 160 > $O/traffic_720p.log 2>&1
bash tools/traffic_pmc.sh davis480p_r50_top20_ref5 --stateful --ref-num 5 --topk 20 > $O/traffic_topk.log 2>&1
for t in davis480p_r50_dense ytvos720p_r50_dense davis480p_r50_top20_ref5; do cp gpurun_out/traffic_$t/traffic.json $O/traffic_$t.json; cp gpurun_out/traffic_$t/summary.txt $O/traffic_$t.txt; done
echo "traffic done"
fi
# 4. SQ counters of the dense kernel
if [[ "$PARTS" == *" 4 "* ]]; then
bash tools/pmc.sh r02 "" --stateful > $O/pmc.log 2>&1
cp gpurun_out/pmc_r02_summary.txt $O/prop_kernel_pmc.txt
fi
echo "all done"
