"""Pointwise-convolution A/B on the GPU box: every 1x1 layer shape of the ResNet-50 encoder at 480p, as (a) MIOpen convolution
(solver search on) + vosprop_bias_act and (b) vosprop_pointwise_conv (hipBLASLt GEMM, epilogue inside); then the whole encoder
both ways.  Usage: python tools/pw_bench.py [--batch 64]"""
import argparse
import importlib
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
vn = importlib.import_module('semi-supervised-vos_amd.vos_net')


def timed(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3   # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--no-find', action='store_true')
    args = ap.parse_args()
    dev = torch.device('cuda', 0)
    dt = torch.bfloat16
    torch.backends.cudnn.benchmark = not args.no_find
    B = args.batch
    # (name, cin, cout, H, W, relu, residual, count per frame)
    shapes = [('l1.conv1', 256, 64, 120, 214, True, False, 2), ('l1.0.conv1', 64, 64, 120, 214, True, False, 1),
              ('l1.conv3', 64, 256, 120, 214, True, True, 3), ('l1.0.ds', 64, 256, 120, 214, False, False, 1),
              ('l2.0.conv1', 256, 128, 120, 214, True, False, 1), ('l2.conv1', 512, 128, 60, 107, True, False, 3),
              ('l2.conv3', 128, 512, 60, 107, True, True, 4),
              ('l3.0.conv1', 512, 256, 60, 107, True, False, 1), ('l3/4.conv1', 1024, 256, 60, 107, True, False, 8),
              ('l3/4.conv3', 256, 1024, 60, 107, True, True, 9), ('l3.0.ds', 512, 1024, 60, 107, False, False, 1),
              ('adjust', 1024, 256, 60, 107, False, False, 1)]
    tot_a = tot_b = 0.0
    print(f'batch {B}: per-call us   conv+bias_act | gemm   (bytes-at-roof us)')
    for name, cin, cout, h, w, relu, res, cnt in shapes:
        conv = torch.nn.Conv2d(cin, cout, 1, bias=True).to(dev).to(dt).to(memory_format=torch.channels_last)
        x = torch.randn(B, cin, h, w, device=dev).to(dt).contiguous(memory_format=torch.channels_last)
        r = torch.randn(B, cout, h, w, device=dev).to(dt).contiguous(memory_format=torch.channels_last) if res else None
        bias = conv.bias.detach()
        with torch.no_grad():
            vn._POINTWISE_GEMM = False
            ta = timed(lambda: vn.conv_bias_act(x, conv, bias, r, relu))
            vn._POINTWISE_GEMM = True
            tb = timed(lambda: vn.conv_bias_act(x, conv, bias, r, relu))
        px = B * h * w
        roof = (px * (cin + cout * (2 if res else 1)) * 2) / 5.0e12 * 1e6
        print(f'{name:12s} {cin:5d}->{cout:5d} {h}x{w}  {ta:9.1f} | {tb:9.1f}   ({roof:7.1f})  x{cnt}', flush=True)
        tot_a += ta * cnt
        tot_b += tb * cnt
        del x, r, conv
    print(f'sum over the 1x1 layers of one batch: {tot_a / 1e3:.2f} ms | {tot_b / 1e3:.2f} ms  = {tot_a / B:.1f} | {tot_b / B:.1f} us/frame')

    net = vn.VOSNet('resnet50')
    net.prepare_for_inference(dev, dt, miopen_find=not args.no_find)
    x = torch.randn(B, 3, 480, 854, device=dev).to(dt).contiguous(memory_format=torch.channels_last)
    for flag in (False, True):
        vn._POINTWISE_GEMM = flag
        g = vn.GraphedEncoder(net)
        with torch.no_grad():
            g(x)
            t = timed(lambda: g(x), iters=5)
        print(f'whole encoder, graphed, batch {B}, pointwise GEMM {flag}: {t / 1e3:.2f} ms = {t / B:.1f} us/frame', flush=True)
    vn._POINTWISE_GEMM = True


if __name__ == '__main__':
    main()
