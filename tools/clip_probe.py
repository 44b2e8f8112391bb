"""How long do the objects of bench.py's synthetic clip survive label propagation on random-init encoder features?  (dev probe)
python tools/clip_probe.py <feature scale> [tint scales]"""
import importlib
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench

vos = importlib.import_module('semi-supervised-vos_amd')
vos_net = importlib.import_module('semi-supervised-vos_amd.vos_net')
dev = torch.device('cuda', 0)
H, W = 480, 854
FS = float(sys.argv[1])      # multiply the features by this (= scale the embedding head's random init)
for scale in [float(a) for a in sys.argv[2:]] or [0.0, 0.5, 1.0]:
    torch.manual_seed(0)
    clip, ann = bench.synthetic_clip(H, W, 24, 0, 'cpu', tint_scale=scale)
    clip = clip.to(dev).to(torch.float16).contiguous(memory_format=torch.channels_last)
    net = vos_net.VOSNet('resnet50')
    net.prepare_for_inference(dev, torch.float16)
    Hd, Wd = vos.feature_map_size(H, W)
    eng = vos.PropagationEngine(Hd, Wd, device=0, ref_num=9)
    eng.begin_video(ann)
    with torch.no_grad():
        f = (net(clip).float() * FS).to(torch.float16)
    print(f'tint {scale}: annotation histogram {np.bincount(ann.reshape(-1), minlength=4).tolist()}  feature norm mean {float(f.float().norm(dim=1).mean()):.2f}')
    for t in range(24):
        _, m = eng.step(f[t][None], want_pred=False, want_mask=True)
        if m is not None and t in (1, 2, 3, 6, 10, 16, 23):
            print(f'   frame {t:2d}: {torch.bincount(m.reshape(-1).long(), minlength=4).tolist()}')
    eng.close()
