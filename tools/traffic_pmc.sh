#!/bin/bash
# HBM-side traffic of the propagation kernel(s) of one workload: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes, as
# MI355X_MICROARCH.md prescribes (plus GRBM_GUI_ACTIVE, TCC hit / miss).  Usage (on the GPU box):
#     bash tools/traffic_pmc.sh <tag> [prop_bench args]        e.g.  bash tools/traffic_pmc.sh davis480p_r50_dense --stateful
# TRAFFIC_CMD="python bench.py --workload ... --steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end" profiles that command instead of
# tools/prop_bench.py (the bench CLIP rather than flat logits; the last 6 dispatches of each kernel are still what is averaged).
# Writes gpurun_out/traffic_<tag>/{summary.txt,traffic.json} (copy the json to profiles/r04_prop_kernel_traffic_<tag>.json;
# it carries the hash of the kernel sources it was taken with - bench.py quotes it only while that hash matches the build).
R=$(cd "$(dirname "$0")/.." && pwd)
tag=$1; shift
O=$R/gpurun_out/traffic_$tag
cd /tmp && export TMPDIR=/tmp
rm -rf $O && mkdir -p $O
i=0
for set in "GRBM_GUI_ACTIVE FETCH_SIZE" "GRBM_GUI_ACTIVE WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  if [ -n "$TRAFFIC_CMD" ]; then
    (cd $R && timeout -k 10 500 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p$i -- $TRAFFIC_CMD > $O/p$i.log 2>&1) || echo "pass $i failed"
  else
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p$i -- python $R/tools/prop_bench.py --iters 5 "$@" > $O/p$i.log 2>&1 || echo "pass $i failed"
  fi
done
python $R/tools/pmc_summary.py $O/p1 $O/p2 $O/p3 | tee $O/summary.txt
python - "$O/summary.txt" "$tag" "${TRAFFIC_CMD:-tools/prop_bench.py --iters 5 $*}" "$R" <<'PY'
import collections, json, re, sys
sys.path.insert(0, sys.argv[4])
import bench
rows = collections.defaultdict(dict)
for ln in open(sys.argv[1]):
    m = re.match(r'(\S+)\s+(\S+)\s+n=\s*\d+ mean=([0-9.e+-]+)', ln)
    if m:
        rows[m.group(1)][m.group(2)] = float(m.group(3))
kern, tot, us = [], 0.0, 0.0
for k, c in rows.items():
    if 'FETCH_SIZE' in c and 'WRITE_SIZE' in c:
        # KiB -> bytes; FETCH_SIZE under-counts 16 B/lane streams by 2 on gfx950 (MI355X_MICROARCH.md, HBM section)
        b = 2 * c['FETCH_SIZE'] * 1024 + c['WRITE_SIZE'] * 1024
        kern.append({'kernel': k, 'fetch_kib': c['FETCH_SIZE'], 'write_kib': c['WRITE_SIZE'], 'traffic_bytes': b,
                     'kernel_us_profiled': c.get('kernel_us(profiled)'),
                     'l2_hit_rate': c['TCC_HIT_sum'] / (c['TCC_HIT_sum'] + c['TCC_MISS_sum']) if 'TCC_HIT_sum' in c else None})
        tot += b
        us += c.get('kernel_us(profiled)') or 0.0
out = {'workload': sys.argv[2], 'kernel_source_hash': bench.kernel_source_hash(), 'bench_args': sys.argv[3], 'kernels': kern,
       'traffic_bytes_per_launch': tot, 'hbm_gb_per_s': tot / us / 1e3 if us else None,
       'how': 'tools/traffic_pmc.sh: rocprofv3 --kernel-trace --pmc, FETCH_SIZE and WRITE_SIZE in separate passes, mean of the last 6 '
              'dispatches (full-N propagations); FETCH_SIZE x2 (gfx950: wide 16 B/lane streams are counted at half, '
              'MI355X_MICROARCH.md HBM section); the propagation kernels of one step summed'}
json.dump(out, open(sys.argv[1].replace('summary.txt', 'traffic.json'), 'w'), indent=1)
print(json.dumps(out)[:600])
PY
