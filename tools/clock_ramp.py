#!/usr/bin/env python3
"""Why do the first propagation launches after an encoder batch run slower?  (round-1 verdict, item 5 iv)

Reads one `rocprofv3 --kernel-trace --pmc <counters> --output-format csv` directory of a bench.py run and prints, for the dense
propagation kernel, the mean duration and the mean of each counter BY POSITION after the last encoder kernel (position 0 = the first
propagation after an encoder batch).  With GRBM_GUI_ACTIVE the effective shader clock of a dispatch is counter / 8 XCDs / duration;
with TCC_HIT_sum / TCC_MISS_sum the L2 hit rate says whether the caches were cold.

    rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d D -- python bench.py --no-cpu-baseline --no-end-to-end
    python tools/clock_ramp.py D [first] [count]      # only propagation launches first .. first+count-1 (the timed region)
"""
import collections
import csv
import glob
import sys

d = sys.argv[1]
kt = glob.glob(f'{d}/*/*_kernel_trace.csv')[0]
cc = glob.glob(f'{d}/*/*_counter_collection.csv')
rows = sorted(csv.DictReader(open(kt)), key=lambda r: int(r['Start_Timestamp']))
counters = collections.defaultdict(lambda: collections.defaultdict(float))
if cc:
    for r in csv.DictReader(open(cc[0])):
        counters[int(r['Dispatch_Id'])][r['Counter_Name']] += float(r['Counter_Value'])
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
count = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 30
pos = 0
seen = 0
by_pos = collections.defaultdict(list)
for r in rows:
    name = r['Kernel_Name']
    dur = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    if 'prop_dense_kernel' in name or 'prop_bf16_kernel' in name:
        if first <= seen < first + count:
            by_pos[pos].append((dur, counters.get(int(r['Dispatch_Id']), {})))
        seen += 1
        pos += 1
    elif 'Cijk' in name or 'conv' in name.lower() or 'igemm' in name:      # an encoder kernel: the next propagation is position 0
        pos = 0
names = sorted({k for v in by_pos.values() for _, c in v for k in c})
print('position  launches  mean_us ' + ' '.join(f'{n:>16s}' for n in names) + ('   clock_GHz' if 'GRBM_GUI_ACTIVE' in names else '')
      + ('  l2_hit' if 'TCC_HIT_sum' in names else ''))
for p in sorted(by_pos)[:64]:
    v = by_pos[p]
    if len(v) < 2:
        continue
    us = sum(x for x, _ in v) / len(v)
    line = f'{p:8d} {len(v):9d} {us:8.1f} '
    means = {}
    for n in names:
        means[n] = sum(c.get(n, 0.0) for _, c in v) / len(v)
        line += f'{means[n]:16.5g} '
    if 'GRBM_GUI_ACTIVE' in means:
        line += f'  {means["GRBM_GUI_ACTIVE"] / 8 / us / 1e3:9.3f}'
    if 'TCC_HIT_sum' in means:
        line += f'  {means["TCC_HIT_sum"] / max(1.0, means["TCC_HIT_sum"] + means["TCC_MISS_sum"]):6.3f}'
    print(line)
