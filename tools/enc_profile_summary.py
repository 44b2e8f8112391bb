"""Per-frame encoder kernel table from the kernel trace of tools/enc_profile.py: only the dispatches of the last graph replays
(the solver search and the warm-up are cut off by taking, per kernel name, the calls that repeat in every replay)."""
import csv
import glob
import sys
from collections import defaultdict

d, B, replays = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
f = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the replays are the tail of the trace: find the launches of one replay by counting from the end
names = [r['Kernel_Name'] for r in rows]
durs = [int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rows]
# period = smallest p such that the last 3p names are 3 repeats
n = len(names)
period = None
for p in range(20, 400):
    if n >= 3 * p and names[n - p:] == names[n - 2 * p:n - p] == names[n - 3 * p:n - 2 * p]:
        period = p
        break
print('launches per replay:', period)
agg = defaultdict(lambda: [0, 0])
use = min(replays - 1, 10)
for i in range(n - use * period, n):
    k = names[i]
    short = k[:90]
    agg[short][0] += 1
    agg[short][1] += durs[i]
tot = sum(v[1] for v in agg.values())
print(f'GPU time per frame: {tot / use / B / 1e3:.1f} us')
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f'{v[1] / use / B / 1e3:8.1f} us/frame  {v[0] // use:4d} launches/batch  {k}')
