#!/bin/bash
# Round-3 evidence in one go (on the GPU box); outputs under gpurun_out/r03/ (copy what is cited into profiles/ as r03_*).
# R03_PARTS="1 3 4" picks parts: 1 kernel stats of the default bench command, 3 HBM traffic of every workload's propagation
# kernel(s) (FETCH_SIZE / WRITE_SIZE in separate passes), 4 SQ counters of the dense kernel and of the top-k passes.
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out/r03
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
PARTS=" ${R03_PARTS:-1 3 4 5 6 7 8} "
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
# 5. which encoder kernels are reproducible (tools/determinism_probe.py), with the convolution kernel names of each mode
if [[ "$PARTS" == *" 5 "* ]]; then
cd /tmp
export VOSPROP_CACHE_DIR=/tmp/vpc_det && mkdir -p $VOSPROP_CACHE_DIR
probe() { python3 $R/tools/determinism_probe.py "$@" 2>&1 | grep -v amdgpu.ids; }
kernels() {   # convolution kernels of one probe run: calls, average us, name
  rm -rf /tmp/detprof; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/detprof -- python3 $R/tools/determinism_probe.py "$@" --repeats 1 > /dev/null 2>&1
  python3 - "$(find /tmp/detprof -name '*kernel_stats.csv' | head -1)" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Name']
    if any(k in n.lower() for k in ('igemm', 'conv', 'winograd', 'naive', 'implicit')) and 'pw_' not in n:
        print(f"   {r['Calls']:>5s} calls {float(r['AverageNs']) / 1e3:9.1f} us  {n[:140]}")
PY
}
{
small="--model resnet18 --size 96 160 --batch 32 --frames 9"
big="--model resnet50 --size 480 854 --batch 32 --frames 12 --repeats 2"
echo "==== A. the failing test's shapes (resnet18 f16, 96x160 frames, batch 32), library defaults: process 1, then the digest line of process 2"
probe $small; probe $small | grep -E "digest|run-to-run"
echo "==== A. convolution kernels (rocprofv3 kernel stats)"; kernels $small
echo "==== B. same shapes, inference.set_deterministic() (this size: MIOpen's ASM implicit-GEMM NHWC family switched off): process 1 / 2"
probe $small --det; probe $small --det | grep -E "digest|run-to-run"
echo "==== B. convolution kernels"; kernels $small --det
echo "==== C. same shapes, torch.backends.cudnn.deterministic instead (MIOPEN_CONVOLUTION_ATTRIB_DETERMINISTIC): reproducible, naive kernel"
probe $small --cudnn-det | grep -E "run-to-run|digest"
echo "==== C. convolution kernels"; kernels $small --cudnn-det
echo "==== D. the bench / CLI shape (resnet50 f16, 480x854, batch 32), library defaults = what set_deterministic() keeps at this size: process 1 / 2"
probe $big; probe $big | grep -E "digest|run-to-run"
echo "==== D. convolution kernels"; kernels $big
} > $O/determinism_probe.txt 2>&1
echo "determinism done"
cd $R
fi
# 6. bench lines of the other workloads
if [[ "$PARTS" == *" 6 "* ]]; then
: > $O/bench_other_workloads.jsonl
for wl in ytvos720p_r50_dense davis480p_r50_top20_ref5 pair240p_r18 ytvos720p_r50_dense_materialised; do
  timeout -k 10 400 python $R/bench.py --no-cpu-baseline --no-end-to-end --workload $wl >> $O/bench_other_workloads.jsonl 2>> $O/bench_other.err || echo "bench $wl failed"
done
timeout -k 10 300 python $R/bench.py --steps 20 --warmup 5 > $O/bench_driver_form.json 2>> $O/bench_other.err || echo "driver-form bench failed"
echo "other workloads done"
fi
# 7. dense kernel: ablations and stamps
if [[ "$PARTS" == *" 7 "* ]]; then
bash tools/dense_ablate.sh "0 1 2 4 16 32 64 23 119" 2>&1 | grep "dense ablate" | cut -c1-70 > $O/dense_kernel_ablations.txt
bash tools/stamp.sh --stateful 2>&1 | tail -9 > $O/dense_kernel_stamps.txt
echo "ablations done"
fi
# 8. the real command line, host to host, with and without --deterministic
if [[ "$PARTS" == *" 8 "* ]]; then
{
timeout -k 10 300 python tools/cli_bench.py --videos 16 --frames 128 --io-workers 8 8 8 --png-workers 2
timeout -k 10 300 python tools/cli_bench.py --videos 16 --frames 128 --io-workers 8 8 --png-workers 2 --extra=--deterministic
timeout -k 10 300 python tools/cli_bench.py --videos 48 --frames 128 --io-workers 8 --png-workers 2
} > $O/cli_end_to_end.txt 2>&1
echo "cli done"
fi
echo "all done"
