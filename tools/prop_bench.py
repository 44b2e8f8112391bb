#!/usr/bin/env python3
"""Kernel-only bench of the propagation step (no encoder): SURVEY.md section 8d's synthetic kernel inputs -
features ~ N(0, 0.25^2) rounded to bf16 (logits ~ N(0,1)), uniform one-hot labels, frame_idx = 20 so both sigma
branches are live.  Prints the mean kernel time from HIP events (vosprop_time_last_propagation).
Used under rocprofv3 for the PMC passes."""
import argparse
import os
import importlib
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--hd', type=int, default=60)
    ap.add_argument('--wd', type=int, default=107)
    ap.add_argument('--ref-num', type=int, default=9)
    ap.add_argument('--d', type=int, default=4)
    ap.add_argument('--iters', type=int, default=50)
    ap.add_argument('--scale', type=float, default=0.25)
    ap.add_argument('--prob', action='store_true')
    ap.add_argument('--topk', type=int, default=0)
    ap.add_argument('--materialise', action='store_true', help='the materialised-affinity (HBM-stress) variant')
    ap.add_argument('--f32', action='store_true', help='the f32 parity path (VOSPROP_PREC_F32)')
    ap.add_argument('--want-pred', action='store_true', help='stateful: the timed step also returns the prediction (denominators kept)')
    ap.add_argument('--stateful', action='store_true', help='time the begin_video/step path (the dense one-hot kernel)')
    args = ap.parse_args()
    vos = importlib.import_module('semi-supervised-vos_amd')
    dev = torch.device('cuda', 0)
    Hd, Wd, fi = args.hd, args.wd, 20
    T = fi + 1
    g = torch.Generator(device='cpu').manual_seed(0)
    feats = (torch.randn(T, 256, Hd, Wd, generator=g) * args.scale).to(torch.float32 if args.f32 else torch.bfloat16).to(dev)
    lab = torch.randint(0, args.d, (T, Hd * Wd), generator=g)
    oh = torch.zeros(args.d, T, Hd * Wd).scatter_(0, lab.unsqueeze(0), 1.0).to(dev)
    eng = vos.PropagationEngine(Hd, Wd, device=0, ref_num=args.ref_num, probability=args.prob, topk=args.topk,
                                materialise=args.materialise, precision=1 if args.f32 else 0)
    if args.stateful:
        import numpy as np
        ann = np.zeros((Hd * 8, Wd * 8), np.uint8)
        ann[: Hd * 4, : Wd * 4] = args.d - 1
        ann[Hd * 4:, Wd * 2: Wd * 6] = 1
        eng.begin_video(ann)
        for t in range(T):      # like the frame loop, the steps only ask for the mask (--want-pred: and the prediction)
            o, m = eng.step(feats[t], want_pred=args.want_pred, want_mask=True)
            out = o if o is not None else (m.float() if m is not None else None)
    else:
        out = eng.predict(feats[:fi], feats[fi], oh[:, :fi], fi, 40, args.ref_num, 1.0, 8.0, 21.0, args.prob)
    torch.cuda.synchronize()
    us = eng.time_last_propagation(args.iters)
    st = eng.last_stats()
    print(json.dumps({'kernel_us': us, 'tflops': st['flops'] / us / 1e6, 'frac_of_2500': st['flops'] / us / 1e6 / 2500,
                      'algorithmic_gb_per_s': st['bytes'] / us / 1e3,
                      'workgroups': st['workgroups'], 'tiles_per_wg': st['tiles_per_wg'], 'n_ref': st['n_ref'],
                      'hw': st['hw'], 'checksum': float(out.sum())}))
    L = vos._native.lib()
    if hasattr(L, 'vosprop_debug_stamps'):     # -DVOSPROP_STAMP diagnostic build
        import ctypes
        import numpy as np
        NS = 16
        n = st['workgroups'] * 8 * NS
        buf = np.zeros(n, dtype=np.uint64)
        L.vosprop_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        got = L.vosprop_debug_stamps(eng._ctx, buf.ctypes.data_as(ctypes.c_void_p), n)
        assert got == n, got
        a = buf.reshape(-1, 8, NS).astype(np.float64) / st['tiles_per_wg']
        names = ['loop', 'ld-issue', 'mfma', 'bar1', 'prefetch', 'max', 'exp', 'labmfma', 'stwrite', 'bar2']
        if not os.environ.get('VOSPROP_DENSE_TWO_BURST') and not args.topk:
            names = ['head', 'gaps0-7', 'gaps8-15', 'check+labmfma', 'tail+prior', 'dmawait', 'barrier', 'seg-prologue', 'seg-close', '-', 'segments', '-', 'pro:record', 'pro:issue', 'pro:wait']
        print('stamps: cycles per tile')
        for g, sl in (('A (waves 0-3)', slice(0, 4)), ('B (waves 4-7)', slice(4, 8))):
            m = a[:, sl].mean((0, 1))
            print(f'  group {g}: ' + ' '.join(f'{nm}={v:.0f}' for nm, v in zip(names, m)) + f'  total={m.sum():.0f}')
        raw = buf.reshape(-1, 8, NS)
        if raw[:, :, 9].any():     # dense kernel: 100 MHz wall clock at the first and after the last instruction of every wave
            t0 = raw[:, :, 9].astype(np.int64)
            t1 = raw[:, :, 11].astype(np.int64)
            z = t0.min()
            st_us, en_us = (t0 - z) / 100.0, (t1 - z) / 100.0
            nseg = raw[:, 0, 10]
            print('  wall clock (us after the first wave started): wave start mean=%.1f max=%.1f; wave end min=%.1f mean=%.1f max=%.1f'
                  % (st_us.mean(), st_us.max(), en_us.min(), en_us.mean(), en_us.max()))
            for k in sorted(set(nseg.tolist())):
                sel = nseg == k
                print('    workgroups with %d segment(s): %d, end mean=%.1f max=%.1f us' % (k, sel.sum(), en_us[sel].mean(), en_us[sel].max()))
        tot = buf.reshape(-1, 8, NS).astype(np.float64)
        print('  per wave, whole kernel (cycles): stamped total mean=%.0f max=%.0f; segment prologues mean=%.0f max=%.0f; segments mean=%.2f max=%.0f'
              % (tot[:, :, :9].sum(2).mean(), tot[:, :, :9].sum(2).max(), tot[:, :, 7].mean(), tot[:, :, 7].max(),
                 tot[:, :, 10].mean(), tot[:, :, 10].max()))
    eng.close()


if __name__ == '__main__':
    main()
