#!/usr/bin/env python3
"""Fixture for the boundary F-measure, computed by an INDEPENDENT brute-force method (no scipy, no distance transform, none of the
product's code): boundary pixels by the textbook definition (a pixel differs from its east / south / south-east neighbour, borders
as in the DAVIS toolkit: last row looks east only, last column looks south only, corner never), tolerance matching by explicit
disk dilation - every boundary pixel stamps every offset (dy, dx) with dy^2 + dx^2 <= r^2 that stays inside the image - and the
precision / recall / F arithmetic of the DAVIS benchmark (reference src/utils/metrics.py:67-121 describes the same quantity; its
own implementation needs scikit-image, which this image does not have, so it could not be run).

    python tests/golden/make_f_fixture.py     ->  tests/golden/f_measure_fixture.npz  (masks + expected values)
"""
from pathlib import Path

import numpy as np


def boundary_bruteforce(m):
    h, w = m.shape
    b = np.zeros((h, w), bool)
    for y in range(h):
        for x in range(w):
            if y == h - 1 and x == w - 1:
                continue
            v = m[y, x]
            if y == h - 1:
                b[y, x] = v != m[y, x + 1]
            elif x == w - 1:
                b[y, x] = v != m[y + 1, x]
            else:
                b[y, x] = (v != m[y, x + 1]) or (v != m[y + 1, x]) or (v != m[y + 1, x + 1])
    return b


def dilate_bruteforce(b, r):
    h, w = b.shape
    out = np.zeros((h, w), bool)
    R = int(np.floor(r))
    offs = [(dy, dx) for dy in range(-R, R + 1) for dx in range(-R, R + 1) if dy * dy + dx * dx <= r * r]
    ys, xs = np.nonzero(b)
    for y, x in zip(ys, xs):
        for dy, dx in offs:
            yy, xx = y + dy, x + dx
            if 0 <= yy < h and 0 <= xx < w:
                out[yy, xx] = True
    return out


def f_bruteforce(fg, gt, void=None, bound_th=0.008):
    fg, gt = fg.astype(bool), gt.astype(bool)
    if void is not None:
        fg, gt = fg & ~void.astype(bool), gt & ~void.astype(bool)
    r = bound_th if bound_th >= 1 else np.ceil(bound_th * np.sqrt(fg.shape[0] ** 2 + fg.shape[1] ** 2))
    fb, gb = boundary_bruteforce(fg), boundary_bruteforce(gt)
    n_f, n_g = int(fb.sum()), int(gb.sum())
    if n_f == 0 and n_g == 0:
        return 1.0
    if n_f == 0 or n_g == 0:
        return 0.0
    prec = (fb & dilate_bruteforce(gb, r)).sum() / n_f
    rec = (gb & dilate_bruteforce(fb, r)).sum() / n_g
    return 0.0 if prec + rec == 0 else float(2 * prec * rec / (prec + rec))


def blobs(rs, h, w, n):
    yy, xx = np.mgrid[0:h, 0:w]
    m = np.zeros((h, w), bool)
    for _ in range(n):
        cy, cx = rs.uniform(0, h), rs.uniform(0, w)
        ry, rx = rs.uniform(3, h / 3), rs.uniform(3, w / 3)
        m |= ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1
    return m


def main():
    rs = np.random.RandomState(2024)
    cases = {}
    for i, (h, w) in enumerate([(48, 64), (60, 107), (33, 41), (96, 160), (25, 25), (120, 213)]):
        gt = blobs(rs, h, w, 3)
        shift = rs.randint(-4, 5, size=2)
        fg = np.roll(np.roll(gt, shift[0], 0), shift[1], 1) ^ (blobs(rs, h, w, 1) & (rs.rand(h, w) < 0.3))
        void = rs.rand(h, w) < 0.05 if i % 2 else None
        th = [0.008, 0.008, 2, 0.02, 0.008, 0.008][i]
        cases[f'gt{i}'], cases[f'fg{i}'] = gt, fg
        if void is not None:
            cases[f'void{i}'] = void
        cases[f'th{i}'] = np.float64(th)
        cases[f'f{i}'] = np.float64(f_bruteforce(fg, gt, void, th))
        cases[f'bmap{i}'] = boundary_bruteforce(fg)
    cases['n'] = np.int64(6)
    out = Path(__file__).resolve().parent / 'f_measure_fixture.npz'
    np.savez_compressed(out, **cases)
    print({k: float(v) for k, v in cases.items() if k.startswith('f') and k[1:].isdigit()})


if __name__ == '__main__':
    main()
