#!/bin/bash
# Round-3 evidence in one go (on the GPU box); outputs under gpurun_out/r03/ (copy what is cited into profiles/ as r03_*).
# R03_PARTS="1 3 4" picks parts: 1 kernel stats of the default bench command, 3 HBM traffic of every workload's propagation
# kernel(s) (FETCH_SIZE / WRITE_SIZE in separate passes), 4 SQ counters of the dense kernel and of the top-k passes.
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out/r03
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
PARTS=" ${R03_PARTS:-1 3 4} "
# 1. kernel stats of the default bench command
if [[ "$PARTS" == *" 1 "* ]]; then
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_prof -- python $R/bench.py > $O/bench_under_rocprofv3.json 2> $O/bench_prof.err || echo "bench profile failed"
cp $(find $O/bench_prof -name '*kernel_stats.csv' | head -1) $O/bench_kernel_stats.csv 2>/dev/null
rm -rf $O/bench_prof
echo "bench profile done"
fi
# 3. traffic of the propagation kernels of every bench workload
cd $R
if [[ "$PARTS" == *" 3 "* ]]; then
bash tools/traffic_pmc.sh davis480p_r50_dense --stateful > $O/traffic_480p.log 2>&1
bash tools/traffic_pmc.sh ytvos720p_r50_dense --stateful --hd 90 --wd 160 > $O/traffic_720p.log 2>&1
bash tools/traffic_pmc.sh davis480p_r50_top20_ref5 --stateful --ref-num 5 --topk 20 > $O/traffic_topk.log 2>&1
bash tools/traffic_pmc.sh pair240p_r18 --stateful --hd 30 --wd 54 > $O/traffic_240p.log 2>&1
bash tools/traffic_pmc.sh ytvos720p_r50_dense_materialised --stateful --hd 90 --wd 160 --materialise > $O/traffic_mat.log 2>&1
for t in davis480p_r50_dense ytvos720p_r50_dense davis480p_r50_top20_ref5 pair240p_r18 ytvos720p_r50_dense_materialised; do cp gpurun_out/traffic_$t/traffic.json $O/traffic_$t.json; cp gpurun_out/traffic_$t/summary.txt $O/traffic_$t.txt; done
echo "traffic done"
fi
# 4. SQ counters of the dense kernel
if [[ "$PARTS" == *" 4 "* ]]; then
bash tools/pmc.sh r03 "" --stateful > $O/pmc.log 2>&1
cp gpurun_out/pmc_r03_summary.txt $O/prop_kernel_pmc.txt
bash tools/pmc.sh r03topk "" --stateful --ref-num 5 --topk 20 > $O/pmc_topk.log 2>&1
cp gpurun_out/pmc_r03topk_summary.txt $O/topk_kernels_pmc.txt
fi
echo "all done"
