#!/usr/bin/env python3
"""Instruction mix per basic block of one kernel in a hipcc -S dump (dev tool)."""
import collections
import re
import sys

s = open(sys.argv[1]).read()
sym = sys.argv[2]
start = s.index(sym + ':')
end = s.index('.Lfunc_end', start)
cnt = collections.Counter()
blocks = []
cur = 'entry'
for ln in s[start:end].split('\n'):
    t = ln.strip()
    if re.match(r'^\.LBB\d+_\d+:', t) or re.match(r'^; %bb\.\d+:', t):
        blocks.append((cur, cnt))
        cur = t
        cnt = collections.Counter()
        continue
    if not t or t.startswith(';') or t.startswith('.'):
        continue
    cnt[t.split()[0]] += 1
blocks.append((cur, cnt))
minn = int(sys.argv[3]) if len(sys.argv) > 3 else 40
for name, c in blocks:
    tot = sum(c.values())
    if tot > minn:
        valu = sum(v for k, v in c.items() if k.startswith('v_') and 'mfma' not in k)
        print(name, 'total', tot, 'valu', valu, dict(c.most_common(30)))
