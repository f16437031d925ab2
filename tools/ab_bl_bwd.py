#!/usr/bin/env python
"""tools/ab_bl_bwd.py -- RGB bilinear-warp backward, A/B in one process (measurement build): the packed-plane kernel on
its tile shapes (product: 64 x 32 sites on 512 lanes) against the fp64-plane kernel of rounds 1-2 (bl_cap 0), three flows,
32x3x720x1280 and 8x3x256x448.  Run on the GPU box."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "memc-net_amd"))
import my_package._ext.my_lib as L      # noqa: E402
from tools import measure as M          # noqa: E402
from tools import synth                 # noqa: E402
from tools.bench_ops import time_launches  # noqa: E402

M.use()
dev = torch.device("cuda:0")
print(M.version(), torch.cuda.get_device_name(0))
for B, H, W, tag in ((32, 720, 1280, "720p"), (8, 256, 448, "c2")):
    for kind in ("smooth", "iid", "video"):
        t = synth.torch_inputs(dev, B, 3, H, W, flow_kind=kind, with_grad=True)
        x, f, g = t["x"], t["flow"], t["gout"]
        g1, g2 = torch.zeros_like(x), torch.zeros_like(f)
        burst = 20 if B * H * W < 4e6 else 1
        row = []
        for rnd in range(2):
            for cap, name in ((-1, "packed, 64x32 / 512 lanes"), (4, "packed, 64x16 / 256 lanes"), (3, "64x16 / 256, 39 KiB"), (0, "fp64 plane per colour")):
                M.set_variant("bl_cap", cap)
                med, mn = time_launches(lambda: L.InterpolationLayer_gpu_backward(x, f, g, g1, g2), lambda: g1.zero_(), burst=burst)
                row.append((name, med))
        M.set_variant("bl_cap", -1)
        sites = B * H * W
        for name in ("packed, 64x32 / 512 lanes", "packed, 64x16 / 256 lanes", "64x16 / 256, 39 KiB", "fp64 plane per colour"):
            best = min(m for n, m in row if n == name)
            print("interpolation_bwd %s C=3 %dx%dx%d flow=%-6s %-27s %8.1f us  %5.1f%% of 8 TB/s" % (
                tag, B, H, W, kind, name, best * 1e6, 100 * sites * 52 / best / 8e12), flush=True)
