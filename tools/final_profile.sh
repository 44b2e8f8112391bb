#!/bin/bash
# Round-end evidence in one go (on the GPU box): rocprofv3 kernel stats of the default bench.py command, the steady-state encoder
# kernel table, the pointwise A/B, the CLI end to end, and the other workloads.  Outputs under gpurun_out/final/.
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out/final
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_prof -- python $R/bench.py > $O/bench_under_rocprofv3.json 2> $O/bench_prof.err || echo "bench profile failed"
cp $(find $O/bench_prof -name '*kernel_stats.csv' | head -1) $O/bench_kernel_stats.csv 2>/dev/null
rm -rf $O/bench_prof
echo "bench profile done"
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/encprof -- python $R/tools/enc_profile.py 64 > $O/enc_profile.log 2>&1 || echo "encoder profile failed"
(grep "^encoder batch" $O/enc_profile.log; python $R/tools/enc_profile_summary.py $O/encprof 64 20) > $O/encoder_kernel_breakdown.txt 2>&1
rm -rf $O/encprof
echo "encoder profile done"
cd $R
timeout -k 10 300 python tools/cli_bench.py --videos 16 --frames 128 --io-workers 8 8 8 --png-workers 2 > $O/cli_end_to_end.txt 2>&1 || echo "cli bench failed"
timeout -k 10 300 python tools/cli_bench.py --videos 16 --frames 128 --io-workers 8 --png-workers 2 --extra "--encoder-batch 64" >> $O/cli_end_to_end.txt 2>&1 || echo "cli bench failed"
timeout -k 10 300 python tools/cli_bench.py --videos 48 --frames 128 --io-workers 8 --png-workers 2 >> $O/cli_end_to_end.txt 2>&1 || echo "cli bench failed"
echo "cli done"
for wl in davis480p_r50_top20_ref5 ytvos720p_r50_dense pair240p_r18; do
  timeout -k 10 300 python bench.py --workload $wl --no-cpu-baseline > $O/bench_$wl.json 2> $O/bench_$wl.err || echo "$wl failed"
done
timeout -k 10 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err || echo "bench failed"
echo "all done"
