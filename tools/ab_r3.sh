#!/bin/bash
# Same-box A/B of this tree's mask-only kernel against round 3's (a copy of commit ee84162 built under build/r3 by
# `git archive ee84162 --prefix=r3/ | tar -x -C build/`): tools/prop_bench.py --stateful, alternating, three rounds.
R=$(cd "$(dirname "$0")/.." && pwd)
for i in 1 2 3; do
  a=$(cd $R && python tools/prop_bench.py --stateful "$@" 2>/dev/null | grep -o '"kernel_us": [0-9.]*' | cut -c14-20)
  b=$(cd $R/build/r3 && python tools/prop_bench.py --stateful "$@" 2>/dev/null | grep -o '"kernel_us": [0-9.]*' | cut -c14-20)
  echo "r4 $a | r3 $b"
done
