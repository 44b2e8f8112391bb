#!/usr/bin/env python3
"""Debug tool: prop_mask_kernel against prop_dense_kernel (denominators kept) on the SAME propagation, partial slot by partial
slot (vosprop_debug_partials).  Both accumulate Y[k, t] = sum_r L[k, r] w 2^((s - m) c) against their own reference level m, so
log2 Y + m c must agree per class.   python tools/dbg_mask.py --hd 12 --wd 20 --frames 3 --ref-num 9"""
import argparse
import ctypes
import importlib
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--hd', type=int, default=12)
    ap.add_argument('--wd', type=int, default=20)
    ap.add_argument('--frames', type=int, default=3)
    ap.add_argument('--ref-num', type=int, default=9)
    ap.add_argument('--scale', type=float, default=0.25)
    ap.add_argument('--d', type=int, default=3)
    ap.add_argument('--verbose', type=int, default=6)
    args = ap.parse_args()
    vos = importlib.import_module('semi-supervised-vos_amd')
    dev = torch.device('cuda', 0)
    Hd, Wd = args.hd, args.wd
    rs = np.random.RandomState(1)
    ann = np.zeros((Hd * 8, Wd * 8), np.uint8)
    ann[: Hd * 4, : Wd * 4] = 1
    ann[Hd * 3:, Wd * 5:] = args.d - 1
    eng = vos.PropagationEngine(Hd, Wd, device=0, ref_num=args.ref_num)
    eng.begin_video(ann)
    for t in range(args.frames):
        f = torch.from_numpy(rs.randn(256, Hd, Wd).astype(np.float32) * args.scale).to(dev)
        eng.step(f, want_pred=True, want_mask=True)
    torch.cuda.synchronize()
    st = eng.last_stats()
    L = vos._native.lib()
    L.vosprop_debug_partials.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
    L.vosprop_debug_partials.restype = ctypes.c_int
    d = int(ann.max()) + 1
    cap = 1 << 24
    out = []
    for which in (0, 1):
        buf = np.zeros(cap, np.float32)
        n = L.vosprop_debug_partials(eng._ctx, which, buf.ctypes.data_as(ctypes.c_void_p), cap)
        assert n > 0, n
        out.append(buf[:n].reshape(-1, 2 + d, 256).copy())
    dense, mask = out
    c = 1.4426950408889634
    print('stats', st, 'slots', dense.shape[0])
    HW = Hd * Wd
    worst = 0.0
    for s in range(dense.shape[0]):
        md, mm = dense[s, 0], mask[s, 0]
        Yd, Ym = dense[s, 2:], mask[s, 2:]
        with np.errstate(divide='ignore', invalid='ignore'):
            ld = np.log2(Yd) + md * c
            lm = np.log2(Ym) + mm * c
        ok = np.isfinite(ld) & np.isfinite(lm) & (Yd > 1e-30)
        diff = np.where(ok, np.abs(ld - lm), 0.0)
        both_zero = (Yd == 0) & (Ym == 0)
        bad_zero = ((Yd > 1e-20 * Yd.max(0, keepdims=True)) & (Ym == 0)) | (~np.isfinite(Ym))
        w = float(diff.max())
        worst = max(worst, w)
        if s < args.verbose or w > 0.05 or bad_zero.any():
            j = int(np.unravel_index(np.argmax(diff), diff.shape)[1])
            print(f'slot {s}: max |log2 Yd + md c - log2 Ym - mm c| = {w:.4f} at col {j}; nonfinite/zero mismatches {int(bad_zero.sum())}; '
                  f'md[{j}]={md[j]:.3f} mm[{j}]={mm[j]:.3f} Yd={Yd[:, j]} Ym={Ym[:, j]}')
    print('worst', worst)
    eng.close()


if __name__ == '__main__':
    main()
