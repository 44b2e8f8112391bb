#!/usr/bin/env python3
"""Where does a propagation step's time go on the GPU?  Reads one `rocprofv3 --kernel-trace --output-format csv -d D` directory of a
bench.py run and prints, over the steady-state propagation steps (a step = everything from one propagation kernel's start to the
next one's start, steps that contain an encoder kernel excluded), the mean duration of every kernel of a step and the mean idle
gap in front of it.

    rocprofv3 --kernel-trace --output-format csv -d D -- python bench.py --no-cpu-baseline --no-end-to-end
    python tools/step_timeline.py D
"""
import collections
import csv
import glob
import sys

d = sys.argv[1]
kt = glob.glob(f'{d}/*/*_kernel_trace.csv')[0]
rows = sorted(csv.DictReader(open(kt)), key=lambda r: int(r['Start_Timestamp']))
ev = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows]
is_prop = lambda n: 'prop_dense_kernel' in n or 'prop_bf16_kernel' in n
starts = [i for i, e in enumerate(ev) if is_prop(e[2])]
short = lambda n: n.split('(')[0].replace('void ', '').replace('vosprop::', '')[:60]
steps = []
for a, b in zip(starts[:-1], starts[1:]):
    seq = ev[a:b]
    if any('Cijk' in n or 'conv' in n.lower() or 'igemm' in n for _, _, n in seq):
        continue
    if ev[a][1] - ev[a][0] < 100000:       # a priming launch (few reference frames)
        continue
    steps.append((seq, ev[b][0]))
shape = collections.Counter(tuple(short(n) for _, _, n in seq) for seq, _ in steps)
sig, cnt = shape.most_common(1)[0]
print(f'{len(steps)} steady-state steps, {cnt} of them with the launch sequence below')
acc = [[0.0, 0.0] for _ in sig]
tail = 0.0
total = 0.0
for seq, nxt in steps:
    if tuple(short(n) for _, _, n in seq) != sig:
        continue
    prev_end = None
    for k, (s, e, n) in enumerate(seq):
        acc[k][0] += (e - s) / 1e3
        if prev_end is not None:
            acc[k][1] += (s - prev_end) / 1e3
        prev_end = e
    tail += (nxt - prev_end) / 1e3
    total += (nxt - seq[0][0]) / 1e3
print(f'{"kernel":62s} {"gap before us":>14s} {"duration us":>12s}')
for k, n in enumerate(sig):
    print(f'{n:62s} {acc[k][1] / cnt:14.2f} {acc[k][0] / cnt:12.2f}')
print(f'{"(gap to the next propagation kernel)":62s} {tail / cnt:14.2f}')
print(f'step total {total / cnt:.2f} us')
