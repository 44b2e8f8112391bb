#!/bin/bash
# Round-4 evidence in one go (on the GPU box); outputs under gpurun_out/r04p/ (copy what is cited into profiles/ as r04_*).
# R04_PARTS="1 3" picks parts: 1 kernel stats (rocprofv3 --kernel-trace --stats) of the driver's bench command, 2 the driver-form
# bench line itself + the other workloads, 3 HBM traffic of every workload's propagation kernel(s) (FETCH_SIZE / WRITE_SIZE in
# separate passes), 4 SQ counters of prop_mask_kernel and of the top-k passes, 5 the CLI host to host.
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out/r04p
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
PARTS=" ${R04_PARTS:-1 2 3 4} "
if [[ "$PARTS" == *" 2 "* ]]; then
cd $R
timeout -k 10 300 python $R/bench.py --steps 20 --warmup 5 > $O/bench_driver_form.json 2> $O/bench_driver_form.err || echo "driver-form bench failed"
: > $O/bench_other_workloads.jsonl
for wl in ytvos720p_r50_dense davis480p_r50_top20_ref5 pair240p_r18 ytvos720p_r50_dense_materialised; do
  timeout -k 10 400 python $R/bench.py --no-cpu-baseline --no-end-to-end --workload $wl >> $O/bench_other_workloads.jsonl 2>> $O/bench_other.err || echo "bench $wl failed"
done
timeout -k 10 400 python $R/bench.py > $O/bench_default.json 2>> $O/bench_other.err || echo "default bench failed"
echo "bench lines done"
cd /tmp
fi
if [[ "$PARTS" == *" 1 "* ]]; then
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_prof -- python $R/bench.py --steps 20 --warmup 5 > $O/bench_under_rocprofv3.json 2> $O/bench_prof.err || echo "bench profile failed"
cp $(find $O/bench_prof -name '*kernel_stats.csv' | head -1) $O/bench_kernel_stats.csv 2>/dev/null
rm -rf $O/bench_prof
echo "bench profile done"
fi
cd $R
if [[ "$PARTS" == *" 3 "* ]]; then
bash tools/traffic_pmc.sh davis480p_r50_dense --stateful > $O/traffic_480p.log 2>&1
bash tools/traffic_pmc.sh ytvos720p_r50_dense --stateful --hd 90 --wd 160 > $O/traffic_720p.log 2>&1
bash tools/traffic_pmc.sh davis480p_r50_top20_ref5 --stateful --ref-num 5 --topk 20 > $O/traffic_topk.log 2>&1
bash tools/traffic_pmc.sh pair240p_r18 --stateful --hd 30 --wd 54 > $O/traffic_240p.log 2>&1
# (top-k on the bench CLIP, not flat logits: the data-dependent pass 2 walks far fewer tiles there)
TRAFFIC_CMD="python bench.py --workload davis480p_r50_top20_ref5 --steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end --no-miopen-find" bash tools/traffic_pmc.sh davis480p_r50_top20_ref5_clip > $O/traffic_topk_clip.log 2>&1
bash tools/traffic_pmc.sh ytvos720p_r50_dense_materialised --stateful --hd 90 --wd 160 --materialise > $O/traffic_mat.log 2>&1
for t in davis480p_r50_dense ytvos720p_r50_dense davis480p_r50_top20_ref5 davis480p_r50_top20_ref5_clip pair240p_r18 ytvos720p_r50_dense_materialised; do cp gpurun_out/traffic_$t/traffic.json $O/traffic_$t.json; cp gpurun_out/traffic_$t/summary.txt $O/traffic_$t.txt; done
echo "traffic done"
fi
if [[ "$PARTS" == *" 4 "* ]]; then
bash tools/pmc.sh r04 "" --stateful > $O/pmc.log 2>&1
cp gpurun_out/pmc_r04_summary.txt $O/prop_kernel_pmc.txt
bash tools/pmc.sh r04hd "" --stateful --hd 90 --wd 160 > $O/pmc_hd.log 2>&1
cp gpurun_out/pmc_r04hd_summary.txt $O/prop_kernel_pmc_720p.txt
bash tools/pmc.sh r04topk "" --stateful --ref-num 5 --topk 20 > $O/pmc_topk.log 2>&1
cp gpurun_out/pmc_r04topk_summary.txt $O/topk_kernels_pmc.txt
echo "pmc done"
fi
if [[ "$PARTS" == *" 5 "* ]]; then
{
timeout -k 10 300 python tools/cli_bench.py --videos 16 --frames 128 --io-workers 8 8 8 --png-workers 2
timeout -k 10 300 python tools/cli_bench.py --videos 16 --frames 128 --io-workers 8 8 --png-workers 2 --extra=--deterministic
} > $O/cli_end_to_end.txt 2>&1
echo "cli done"
fi
echo "all done"
