"""Steady-state encoder only (for rocprofv3 --kernel-trace --stats): ResNet-50 at 480p, batch 64, graphed, 20 replays after
the solver search and the capture.  Usage: rocprofv3 --kernel-trace --stats --output-format csv -d OUT -- python tools/enc_profile.py [batch] [f16|bf16]"""
import importlib
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
vn = importlib.import_module('semi-supervised-vos_amd.vos_net')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
DT = {'bf16': torch.bfloat16, 'f16': torch.float16}[sys.argv[2] if len(sys.argv) > 2 else 'f16']
REPLAYS = 20
dev = torch.device('cuda', 0)
net = vn.VOSNet('resnet50')
net.prepare_for_inference(dev, DT, miopen_find=True, feature_dtype=torch.bfloat16)
x = torch.randn(B, 3, 480, 854, device=dev).to(DT).contiguous(memory_format=torch.channels_last)
g = vn.GraphedEncoder(net)
with torch.no_grad():
    g(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPLAYS):
        g(x)
    e1.record()
    torch.cuda.synchronize()
print(f'encoder batch {B}: {e0.elapsed_time(e1) / REPLAYS:.2f} ms per batch = {e0.elapsed_time(e1) / REPLAYS / B * 1e3:.1f} us/frame')
