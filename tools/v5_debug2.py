"""stateful (v5 kernel) vs stateless (8-wave kernel) on the same inputs (dev tool)"""
import importlib, sys, numpy as np, torch
from pathlib import Path
R = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(R))
vos = importlib.import_module('semi-supervised-vos_amd')
dev = torch.device('cuda', 0)
for (Hd, Wd) in [(8, 8), (8, 16), (12, 20), (16, 16), (30, 54)]:
    HW = Hd * Wd
    g = torch.Generator().manual_seed(1)
    T = 12
    feats = (torch.randn(T, 256, Hd, Wd, generator=g) * 0.25).to(torch.bfloat16).to(dev)
    ann = np.zeros((Hd * 8, Wd * 8), np.uint8); ann[: Hd * 4] = 1; ann[:, : Wd * 3] = 2
    e1 = vos.PropagationEngine(Hd, Wd, device=0)
    e2 = vos.PropagationEngine(Hd, Wd, device=0)
    d = e1.begin_video(ann)
    labs = []
    out = []
    for t in range(T):
        p, m = e1.step(feats[t])
        if t == 0:
            cls0 = torch.from_numpy(ann[::8, ::8].reshape(-1).astype(np.int64))
            labs.append(torch.zeros(d, HW).scatter_(0, cls0[None], 1.0))
            continue
        lab_hist = torch.stack(labs, 1).to(dev)          # (d, t, HW)
        q = e2.predict(feats[:t].float(), feats[t].float(), lab_hist, t, 40, 9, 1.0, 8.0, 21.0, False)
        err = (p - q).abs()
        bad = torch.unique(torch.where(err > 2e-3)[1]).tolist()
        st = e1.last_stats()
        out.append(f't={t} wg={st["workgroups"]} err={err.max().item():.4f} badcols={len(bad)}:{bad[:10]}')
        labs.append(torch.zeros(d, HW).scatter_(0, p.argmax(0).cpu()[None], 1.0))
    print((Hd, Wd), ' | '.join(out))
