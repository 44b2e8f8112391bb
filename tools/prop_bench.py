#!/usr/bin/env python3
"""Kernel-only bench of the propagation step (no encoder): SURVEY.md section 8d's synthetic kernel inputs -
features ~ N(0, 0.25^2) rounded to bf16 (logits ~ N(0,1)), uniform one-hot labels, frame_idx = 20 so both sigma
branches are live.  Prints the mean kernel time from HIP events (vosprop_time_last_propagation).
Used under rocprofv3 for the PMC passes."""
import argparse
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
    eng.close()


if __name__ == '__main__':
    main()
