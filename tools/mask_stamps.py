#!/usr/bin/env python3
"""Where a launch of prop_mask_kernel spends its time OUTSIDE the tile loop: wall-clock stamps (100 MHz s_memrealtime) of every
workgroup's phases, through the debug hook vosprop_debug_mask_stamps.   python tools/mask_stamps.py [--hd 60 --wd 107]"""
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
    ap.add_argument('--hd', type=int, default=60)
    ap.add_argument('--wd', type=int, default=107)
    args = ap.parse_args()
    vos = importlib.import_module('semi-supervised-vos_amd')
    dev = torch.device('cuda', 0)
    Hd, Wd, T = args.hd, args.wd, 21
    g = torch.Generator(device='cpu').manual_seed(0)
    feats = (torch.randn(T, 256, Hd, Wd, generator=g) * 0.25).to(torch.bfloat16).to(dev)
    ann = np.zeros((Hd * 8, Wd * 8), np.uint8)
    ann[: Hd * 4, : Wd * 4] = 3
    ann[Hd * 4:, Wd * 2: Wd * 6] = 1
    eng = vos.PropagationEngine(Hd, Wd, device=0, ref_num=9)
    eng.begin_video(ann)
    for t in range(T):
        eng.step(feats[t], want_pred=False, want_mask=True)
    torch.cuda.synchronize()
    L = vos._native.lib()
    L.vosprop_debug_mask_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    L.vosprop_debug_mask_stamps.restype = ctypes.c_int
    st = eng.last_stats()
    buf = np.zeros(st['workgroups'] * 8, np.uint64)
    n = L.vosprop_debug_mask_stamps(eng._ctx, buf.ctypes.data_as(ctypes.c_void_p), buf.size)
    assert n == st['workgroups'], n
    s = buf.reshape(-1, 8).astype(np.int64)
    z = s[:, 0].min()
    us = (s - z) / 100.0
    names = ['entry', 'segment record*', 'prologue issued', 'prologue landed', 'loop inputs ready', 'first loop done', 'last partial stored', 'exit']
    print(f'{Hd}x{Wd}: {st["workgroups"]} workgroups, {st["tiles_per_wg"]} tile steps each; us after the first workgroup entered the kernel')
    for k, nm in enumerate(names):
        print(f'  {k} {nm:22s} min {us[:, k].min():7.2f}  mean {us[:, k].mean():7.2f}  max {us[:, k].max():7.2f}')
    # by the number of segments a workgroup walks (the work plan, engine.hip build_segments)
    TT = (Hd * Wd + 255) // 256
    NT = st['n_ref'] * ((Hd * Wd + 31) // 32)
    L.vosprop_debug_plan.restype = ctypes.c_int
    L.vosprop_debug_plan.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.c_int]
    la = -(-(Hd * Wd - (TT - 1) * 256) // 32)      # waves of the last target tile that hold a column of the map
    la = la if la < 8 else 0
    nrow = L.vosprop_debug_plan(TT, NT, None, 0, la)
    pb = (ctypes.c_int * (4 * nrow))()
    L.vosprop_debug_plan(TT, NT, pb, nrow, la)
    rows = np.ctypeslib.as_array(pb).reshape(nrow, 4)
    nseg = np.bincount(rows[:, 0], minlength=st['workgroups'])
    steps = np.bincount(rows[:, 0], weights=rows[:, 3], minlength=st['workgroups'])
    for k in sorted(set(nseg.tolist())):
        sel = nseg == k
        print(f'  workgroups with {k} segment(s): {int(sel.sum()):3d}, tile steps {steps[sel].mean():6.1f}, exit mean {us[sel, 7].mean():7.2f} max {us[sel, 7].max():7.2f} us')
    for x in range(8):
        sel = (np.arange(st['workgroups']) % 8) == x
        print(f'  XCD slot {x}: exit mean {us[sel, 7].mean():7.2f} max {us[sel, 7].max():7.2f}')
    # one-segment workgroups by target tile: the LAST tile's waves beyond the map run the staging-only form (prop_mask.h)
    first_tt = np.full(st['workgroups'], -1)
    for b, tt, _, _ in rows[::-1]:
        first_tt[b] = tt
    one = nseg == 1
    for name, sel in (('last target tile', one & (first_tt == TT - 1)), ('other target tiles', one & (first_tt != TT - 1))):
        if sel.any():
            loop = us[sel, 5] - us[sel, 4]
            print(f'  one-segment workgroups on the {name}: {int(sel.sum()):3d}, tile steps {steps[sel].mean():6.1f}, loop {loop.mean():7.2f} us '
                  f'= {1e3 * (loop / steps[sel]).mean():6.1f} ns per step, exit mean {us[sel, 7].mean():7.2f}')
    # (* stamp 1 is rewritten by every segment: it is the LAST segment's record; stamps 2-5 are the first segment's.  Phase lengths
    # are therefore taken over the one-segment workgroups)
    d = np.diff(us[one], axis=1)
    print('  phase lengths, one-segment workgroups (mean us): ' + ', '.join(f'{names[k]}->{names[k + 1]} {d[:, k].mean():.2f}' for k in range(7)))
    eng.close()


if __name__ == '__main__':
    main()
