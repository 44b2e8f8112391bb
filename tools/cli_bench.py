#!/usr/bin/env python3
"""End-to-end CLI throughput (dev tool): synthesise a DAVIS-layout dataset of JPEG frames (480p by default), then time
`main.py inference` with different host I/O settings.  Frames/s here include JPEG decode, H2D, encoder, propagation, D2H and PNG
encoding - the number a user of the command line sees.
    python tools/cli_bench.py --videos 4 --frames 80 --io-workers 1 8 --png-workers 1 2"""
import argparse
import json
import subprocess
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent


def make_dataset(root, videos, frames, H, W, vary=False):
    from PIL import Image
    yy, xx = np.mgrid[0:H, 0:W]
    for v in range(videos):
        rs = np.random.RandomState(v)
        (root / 'JPEGImages' / '480p' / f'v{v:02d}').mkdir(parents=True)
        (root / 'Annotations' / '480p' / f'v{v:02d}').mkdir(parents=True)
        base = rs.randint(0, 255, (H // 16, W // 16, 3)).astype(np.float32)
        for i in range(frames - ((v * 7) % 32 if vary else 0)):     # vary: DAVIS-like, every video its own length
            base = np.clip(base + rs.randn(*base.shape) * 5, 0, 255)
            img = Image.fromarray(base.astype(np.uint8)).resize((W, H), Image.BILINEAR)
            img.save(root / 'JPEGImages' / '480p' / f'v{v:02d}' / f'{i:05d}.jpg', quality=90)
        m = np.zeros((H, W), np.uint8)
        for k in (1, 2, 3):
            cy, cx = rs.randint(H // 4, 3 * H // 4), rs.randint(W // 4, 3 * W // 4)
            m[((yy - cy) / (H / 8)) ** 2 + ((xx - cx) / (W / 8)) ** 2 <= 1] = k
        im = Image.fromarray(m, mode='P')
        im.putpalette([0, 0, 0, 128, 0, 0, 0, 128, 0, 128, 128, 0] + [0] * 756)
        im.save(root / 'Annotations' / '480p' / f'v{v:02d}' / '00000.png')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--videos', type=int, default=4)
    ap.add_argument('--frames', type=int, default=80)
    ap.add_argument('--size', type=int, nargs=2, default=[480, 854])
    ap.add_argument('--model', default='resnet50')
    ap.add_argument('--io-workers', type=int, nargs='+', default=[1, 8])
    ap.add_argument('--png-workers', type=int, nargs='+', default=[2])
    ap.add_argument('--vary', action='store_true', help='videos of different lengths (frames - (7 v mod 32))')
    ap.add_argument('--extra', default='', help='further flags for main.py inference, e.g. "--encoder-batch 64 --miopen-find"')
    ap.add_argument('--profile', action='store_true', help='re-run the last configuration under cProfile and print the top functions')
    args = ap.parse_args()
    import importlib
    import torch
    sys.path.insert(0, str(ROOT))
    vn = importlib.import_module('semi-supervised-vos_amd.vos_net')
    with tempfile.TemporaryDirectory() as td:
        td = Path(td)
        make_dataset(td / 'data', args.videos, args.frames, *args.size, vary=args.vary)
        torch.manual_seed(0)
        torch.save({'state_dict': vn.VOSNet(args.model).state_dict()}, td / 'ckpt.pth.tar')
        for io in args.io_workers:
            for png in args.png_workers:
                t0 = time.perf_counter()
                out = subprocess.run([sys.executable, 'main.py', 'inference', '-d', str(td / 'data'), '-r', str(td / 'ckpt.pth.tar'),
                                      '-m', args.model, '-s', str(td / f'out_{io}_{png}'), '--io-workers', str(io),
                                      '--png-workers', str(png)] + args.extra.split(), cwd=ROOT, capture_output=True, text=True, timeout=1200)
                wall = time.perf_counter() - t0
                if out.returncode != 0:
                    print(json.dumps({'io_workers': io, 'png_workers': png, 'error': out.stderr[-400:]}))
                    continue
                st = json.loads([l for l in out.stdout.splitlines() if l.startswith('{"vosprop_stats"')][0])['vosprop_stats']
                n_png = len(list((td / f'out_{io}_{png}').glob('*/*.png')))
                print(json.dumps({'io_workers': io, 'png_workers': png, 'extra': args.extra, 'frames': st['frames'], 'loop_seconds': round(st['seconds'], 3),
                                  'loop_frames_per_s': round(st['frames'] / st['seconds'], 1), 'process_wall_s': round(wall, 2),
                                  'pngs_written': n_png}), flush=True)
        if args.profile:
            io, png = args.io_workers[-1], args.png_workers[-1]
            subprocess.run([sys.executable, '-m', 'cProfile', '-o', str(td / 'prof.out'), 'main.py', 'inference', '-d', str(td / 'data'),
                            '-r', str(td / 'ckpt.pth.tar'), '-m', args.model, '-s', str(td / 'out_prof'), '--io-workers', str(io),
                            '--png-workers', str(png)], cwd=ROOT, capture_output=True, text=True, timeout=1200)
            import pstats
            st_ = pstats.Stats(str(td / 'prof.out'))
            st_.sort_stats('tottime').print_stats(14)
            st_.sort_stats('cumulative').print_stats('inference_utils|engine.py|io_pipeline|utils.py', 25)


if __name__ == '__main__':
    main()
