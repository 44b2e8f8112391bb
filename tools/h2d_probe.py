import importlib, sys, tempfile, time
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'tools'))
from cli_bench import make_dataset
ds_mod = importlib.import_module('semi-supervised-vos_amd.datasets')
io = importlib.import_module('semi-supervised-vos_amd.io_pipeline')
dev = torch.device('cuda', 0)
with tempfile.TemporaryDirectory() as td:
    td = Path(td)
    make_dataset(td / 'data', 2, 64, 480, 854)
    ds = ds_mod.InferenceDataset(td / 'data' / 'JPEGImages' / '480p', raw_uint8=True)
    torch.zeros(1, device=dev)
    for workers, pin in ((4, True), (4, False), (0, False)):
        loader = io.make_loader(ds, workers, pin=pin)
        xd = torch.empty((16, 480, 854, 3), dtype=torch.uint8, device=dev)
        tc, n, pinned = 0.0, 0, 0
        items = []
        for x, _ in loader:
            items.append(x)
            if len(items) == 16:
                time.sleep(0.02)          # let the GPU side be "slow": the loader queue is full
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i, t in enumerate(items):
                    xd[i:i + 1].copy_(t, non_blocking=True)
                torch.cuda.synchronize()
                tc += time.perf_counter() - t0
                pinned += sum(int(t.is_pinned()) for t in items)
                n += 16
                items.clear()
        print(f'workers={workers} pin={pin}: copy {tc / n * 1e3:.3f} ms/frame, pinned {pinned}/{n}, shape {tuple(x.shape)} {x.dtype}')
