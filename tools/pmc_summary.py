#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc csv output dirs: per propagation kernel, the mean counter value over its LAST `--last` dispatches
(tools/prop_bench.py ends with iters + 1 back-to-back re-runs of the full-N propagation: --last 6 for --iters 5)."""
import argparse
import collections
import csv
import glob
import re

ap = argparse.ArgumentParser()
ap.add_argument('dirs', nargs='+')
ap.add_argument('--last', type=int, default=6)
args = ap.parse_args()


def short(name):
    m = re.search(r'(prop_\w+?_kernel)<([^>]*)>', name)
    return f'{m.group(1)}<{m.group(2).replace(" ", "")}>' if m else name.split('(')[0]


for d in args.dirs:
    fs = glob.glob(f'{d}/*/*_counter_collection.csv')
    if not fs:
        continue
    per = collections.defaultdict(lambda: collections.defaultdict(list))        # kernel -> counter -> [(dispatch id, value)]
    for r in csv.DictReader(open(fs[0])):
        if 'prop_' in r['Kernel_Name'] or 'topk_select' in r['Kernel_Name']:      # what vosprop_time_last_propagation times
            per[short(r['Kernel_Name'])][r['Counter_Name']].append((int(r['Dispatch_Id']), float(r['Counter_Value'])))
    for kern, counters in sorted(per.items()):
        for k, v in sorted(counters.items()):
            # a counter row appears once per dispatch and XCC/SE instance: sum the instances of a dispatch, then take the last ones
            byd = collections.defaultdict(float)
            for did, val in v:
                byd[did] += val
            vals = [byd[k2] for k2 in sorted(byd)][-args.last:]
            print(f'{kern:44s} {k:28s} n={len(vals):2d} mean={sum(vals) / len(vals):.6g}')
    kt = glob.glob(f'{d}/*/*_kernel_trace.csv')
    if kt:
        ds = collections.defaultdict(list)
        for r in csv.DictReader(open(kt[0])):
            if 'prop_' in r['Kernel_Name'] or 'topk_select' in r['Kernel_Name']:      # what vosprop_time_last_propagation times
                ds[short(r['Kernel_Name'])].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
        for kern, v in sorted(ds.items()):
            v = v[-args.last:]
            print(f'{kern:44s} {"kernel_us(profiled)":28s} n={len(v):2d} mean={sum(v) / len(v):.2f}')
