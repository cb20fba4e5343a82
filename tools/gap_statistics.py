#!/usr/bin/env python3
"""Column-gap statistics of rows of the headline matrix, built by the CPU oracle (test infrastructure: tools/ may use oracle/): what a
delta-coded column stream would cost in pad entries and bytes, per (row, tile) segment, for several delta widths and per-tile mixes
with the 12-bit absolute slots; and the product-time model of tools/read_bw_probe.hip's record-size sweep.
  python tools/gap_statistics.py            (about 40 s of CPU: six rows of 9.96e6 cells)  -> profiles/r03_gap_statistics.txt"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as orc  # noqa: E402

tfx = importlib.import_module("tomofast-x_amd")
nx, ny, nz, ox, oy, TC = 256, 256, 152, 316, 316, 4096
N = nx * ny * nz
grid = tfx.synthetic.grid(nx, ny, nz)
xs, ys, zs = tfx.synthetic.observations(nx, ny, ox, oy)
cw = orc.column_weight_type1(grid) * 4e3
K = int(0.02 * N)
sz, pads = [], {w: [] for w in (4, 5, 6, 7, 8)}
hist = np.zeros(14, np.int64)
for r in (0, 15000, 33333, 50000, 70001, 99855):
    c = np.asarray(orc.build_row_grav(grid, (nx, ny, nz), cw, (xs[r], ys[r], zs[r]), 2, K)[0]).astype(np.int64)
    t, lc = c // TC, c % TC
    first = np.r_[True, t[1:] != t[:-1]]
    g = np.where(first, lc, np.r_[0, np.diff(c)])
    ids = np.cumsum(first) - 1
    sz.append(np.bincount(ids))
    hist += np.bincount(np.clip(np.ceil(np.log2(np.maximum(g, 1))).astype(int), 0, 13), minlength=14)
    for w in pads:
        M = (1 << w) - 1
        pads[w].append(np.bincount(ids, weights=np.maximum(0, (g + M - 1) // M - 1)))
sz = np.concatenate(sz)
print("entries %d, (row, tile) segments %d, mean segment %.1f" % (sz.sum(), sz.size, sz.mean()))
print("share of gaps <= 1, 2, 4, ... :", (hist / hist.sum()).round(4))
c12 = sz * 5.625
for w, bpe in ((6, 4.125 + (12 + 7 * 6) / 64), (7, 5.125), (8, 5.25)):
    p = np.concatenate(pads[w])
    cw_ = (sz + p) * bpe
    mix = np.minimum(c12, cw_)
    print("%d-bit deltas: pads %.2f %%, alone %.4f B per entry, per-segment mix with 12-bit slots %.4f (delta share %.3f)" %
          (w, 100 * p.sum() / sz.sum(), cw_.sum() / sz.sum(), mix.sum() / sz.sum(), sz[cw_ < c12].sum() / sz.sum()))
# product-time model: ms per 16.8 M chunks from the probe's sweep (12 B / 9 B / 8 B of slots per lane)
t12, t9, t8 = 6.98, 6.76, 6.58
for w, bpe, tw in ((7, 5.125, t8), (8, 5.25, t9)):
    p = np.concatenate(pads[w])
    use = (sz + p) * bpe < c12
    print("time model, %d-bit mix: %.4f of the all-12-bit time" % (w, (np.where(use, 0, sz).sum() * t12 + np.where(use, sz + p, 0).sum() * tw) / (sz.sum() * t12)))
