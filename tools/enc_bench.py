#!/usr/bin/env python3
"""Encoder throughput variants at 480p (dev tool): eager vs folded BN vs batch vs HIP graph."""
import importlib
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
vn = importlib.import_module('semi-supervised-vos_amd.vos_net')
dev = torch.device('cuda', 0)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n


for model in ('resnet50',):
    for fold in (False, True):
        for B in (1, 4, 8, 16):
            torch.manual_seed(0)
            net = vn.VOSNet(model).prepare_for_inference(dev, torch.bfloat16, fold_bn=fold)
            x = torch.randn(B, 3, 480, 854, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
            with torch.no_grad():
                t = timeit(lambda: net(x))
                line = f'{model} fold={fold} B={B}: eager {t / B * 1e3:.3f} ms/frame'
                try:
                    g = torch.cuda.CUDAGraph()
                    s = torch.cuda.Stream()
                    with torch.cuda.stream(s):
                        for _ in range(2):
                            net(x)
                    torch.cuda.current_stream().wait_stream(s)
                    with torch.cuda.graph(g):
                        y = net(x)
                    tg = timeit(g.replay)
                    line += f', graph {tg / B * 1e3:.3f} ms/frame'
                except Exception as e:  # noqa: BLE001
                    line += f', graph failed: {type(e).__name__}'
            print(line, flush=True)
