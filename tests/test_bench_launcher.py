"""`python bench.py --gpus N` must start its own ranks (round-1 finding: it asserted WORLD_SIZE == --gpus and never launched
anything).  Driven here on the CPU with the stub workload under gloo: the parent never touches a GPU, the children rendezvous on
127.0.0.1, rank 0 prints the single JSON line, the wall time is the max over ranks.  The real workload uses the same protocol with
RCCL (reference: videos are independent, src/utils/inference_utils.py:28-48 - no data-path collective)."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def _run(args, env=None, timeout=300):
    e = dict(os.environ)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, str(ROOT / 'bench.py')] + args, env=e, cwd=str(ROOT), capture_output=True, text=True,
                          timeout=timeout)


@pytest.mark.parametrize('n', [1, 2, 3])
def test_bench_launches_its_own_ranks(n):
    r = _run(['--gpus', str(n), '--steps', '50', '--warmup', '5', '--workload', 'stub_cpu'])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]      # (gloo itself chats on stdout)
    assert len(lines) == 1, r.stdout                       # ONE JSON line, from rank 0
    out = json.loads(lines[0])
    assert out['n_gpus'] == n and out['steps'] == 50 and out['warmup'] == 5 and out['scaling'] == 'weak'
    assert out['value'] > 0 and abs(out['value'] - n * 50 / (out['ms_per_step'] * 50 / 1e3)) < 1e-6 * out['value']


def test_bench_under_an_external_launcher_is_unchanged():
    """The driver's form: torch.distributed.run around `bench.py --gpus 2` (ranks come from the environment, nothing is spawned)."""
    e = dict(os.environ)
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr',
                        '127.0.0.1', '--master-port', '29713', str(ROOT / 'bench.py'), '--gpus', '2', '--steps', '20', '--warmup',
                        '2', '--workload', 'stub_cpu'], env=e, cwd=str(ROOT), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1 and json.loads(lines[0])['n_gpus'] == 2


def test_world_size_mismatch_is_a_clear_error():
    r = _run(['--gpus', '2', '--steps', '5', '--warmup', '1', '--workload', 'stub_cpu'], env={'WORLD_SIZE': '1', 'RANK': '0'})
    assert r.returncode != 0 and 'WORLD_SIZE=1' in (r.stderr + r.stdout)


def test_a_failing_rank_fails_the_launch():
    script = ROOT / 'gpurun_out' / '_rank_exit.py'
    script.parent.mkdir(exist_ok=True)
    script.write_text('import os, sys\nsys.exit(3 if os.environ["RANK"] == "1" else 0)\n')
    code = ("import sys\nsys.path.insert(0, %r)\nimport bench\nsys.exit(bench.launch_ranks(2, [], script=%r))\n"
            % (str(ROOT), str(script)))
    r = subprocess.run([sys.executable, '-c', code], cwd=str(ROOT), capture_output=True, text=True, timeout=120)
    script.unlink()
    assert r.returncode == 3, (r.returncode, r.stderr[-500:])
