#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc csv output dirs: mean counter value per dispatch of the propagation kernel."""
import collections
import csv
import glob
import sys

for d in sys.argv[1:]:
    fs = glob.glob(f'{d}/*/*_counter_collection.csv')
    if not fs:
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if 'prop_' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in sorted(agg.items()):
        print(f'{k:32s} n={len(v):2d} mean={sum(v) / len(v):.5g}')
    kt = glob.glob(f'{d}/*/*_kernel_trace.csv')
    if kt:
        ds = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in csv.DictReader(open(kt[0]))
              if 'prop_' in r['Kernel_Name']]
        if ds:
            print(f'{"kernel_us(profiled)":32s} n={len(ds):2d} mean={sum(ds) / len(ds):.2f}')
