#!/usr/bin/env python
"""tools/probes/box_amplification.py -- CPU only.  How many source pixels does a tile of the adaptive warp have to stage per
output site, as a function of the tile's SHAPE, on the benchmark's flow?  A tile's box is the bounding box of its sites'
4 x 4 windows under the tile's own flow (memc_tile.hpp: tile_bbox), columns rounded to multiples of four as the kernels do.
The 64-channel context warp reads its image 2.0 times (PMC, rounds 1-3); this asks whether ANY tile shape would read it less.

    python tools/probes/box_amplification.py [smooth|video|iid]      -> profiles/r04_box_amplification.txt (by the caller)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tools import synth  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "smooth"
B, H, W = 4, 720, 1280
rng = np.random.default_rng(1234)
flow = synth.np_flow(rng, B, H, W, kind)
fx, fy = flow[:, 0], flow[:, 1]
ys, xs = np.mgrid[0:H, 0:W]
x2, y2 = xs[None] + fx, ys[None] + fy
valid = (x2 >= 0) & (y2 >= 0) & (x2 <= W - 1) & (y2 <= H - 1) & (np.abs(fx) < W / 2) & (np.abs(fy) < H / 2)
ix, iy = np.floor(np.where(valid, x2, 0)).astype(int), np.floor(np.where(valid, y2, 0)).astype(int)
cmin, cmax = np.clip(ix - 1, 0, W - 1), np.clip(ix + 2, 0, W - 1)
rmin, rmax = np.clip(iy - 1, 0, H - 1), np.clip(iy + 2, 0, H - 1)
BIG = 1 << 30
cmin, rmin = np.where(valid, cmin, BIG), np.where(valid, rmin, BIG)
cmax, rmax = np.where(valid, cmax, -1), np.where(valid, rmax, -1)
print("flow=%s, %dx%dx%d: staged source pixels per output site by tile shape (4 x 4 window, box columns rounded to 4)" % (kind, B, H, W))
print("%-12s %10s %12s %12s %12s" % ("tile w x h", "box/site", "mean box w", "mean box h", "fits 3072?"))
for tw, th in ((64, 16), (64, 32), (64, 8), (32, 32), (128, 8), (128, 16), (32, 16), (16, 64), (256, 4), (64, 64)):
    tot_box = tot_sites = 0
    bws, bhs, fit = [], [], []
    for b in range(B):
        for y0 in range(0, H, th):
            for x0 in range(0, W, tw):
                sl = (b, slice(y0, y0 + th), slice(x0, x0 + tw))
                c0, c1, r0, r1 = cmin[sl].min(), cmax[sl].max(), rmin[sl].min(), rmax[sl].max()
                n = valid[sl].size
                if c1 < 0:
                    tot_sites += n
                    continue
                bw = (c1 | 3) + 1 - (c0 & ~3)
                bh = r1 + 1 - r0
                tot_box += bw * bh
                tot_sites += n
                bws.append(bw); bhs.append(bh)
                pitch = max((bw + 15) & ~15, 16)
                fit.append(bh * pitch <= (3072 if tw * th <= 1024 else 3072 * tw * th // 1024))
    print("%4d x %-5d %10.2f %12.1f %12.1f %11.0f%%" % (tw, th, tot_box / tot_sites, np.mean(bws), np.mean(bhs), 100 * np.mean(fit)))
