#!/bin/bash
# kernel-time breakdown of one CLI run (dev tool): bash tools/cli_rocprof.sh
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
D=/tmp/clidata
rm -rf $D gpurun_out/cli_prof && mkdir -p gpurun_out/cli_prof
python - <<'PY'
import sys, importlib, torch
from pathlib import Path
sys.path.insert(0, 'tools'); sys.path.insert(0, '.')
from cli_bench import make_dataset
make_dataset(Path('/tmp/clidata/data'), 8, 100, 480, 854)
vn = importlib.import_module('semi-supervised-vos_amd.vos_net')
torch.manual_seed(0)
torch.save({'state_dict': vn.VOSNet('resnet50').state_dict()}, '/tmp/clidata/ckpt.pth.tar')
PY
python main.py inference -d $D/data -r $D/ckpt.pth.tar -s $D/out0 --io-workers 4 > /dev/null 2>&1   # warm MIOpen's find db
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/cli_prof -o cli -- python main.py inference -d $D/data -r $D/ckpt.pth.tar -s $D/out1 --io-workers 4 > gpurun_out/cli_prof/stdout.log 2>&1
tail -2 gpurun_out/cli_prof/stdout.log
f=$(find gpurun_out/cli_prof -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f'total kernel time {tot/1e6:.1f} ms over {sum(int(r["Calls"]) for r in rows)} launches')
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:14]:
    print(f"{float(r['TotalDurationNs'])/1e6:9.1f} ms {int(r['Calls']):6d} calls {float(r['AverageNs'])/1e3:9.1f} us  {r['Name'][:90]}")
PY
