#!/usr/bin/env python
"""tools/probes/proj_pan_phases.py -- where does the owner kernel's time go under a camera pan with hole filling?  The
timestamp instance of proj_owner5 (measurement build, variant -41): per workgroup the clocks of every phase, as
percentiles over the workgroups, for the benchmark's flow and with a pan of 40 px on top (the uncovered bands are 9 % of the
tiles: their workgroups are the upper percentiles)."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "memc-net_amd"))
import my_package._ext.my_lib as L      # noqa: E402
from tools import measure as M  # noqa: E402
M.use()
from tools import synth                 # noqa: E402

dev = torch.device("cuda:0")
B, H, W, TH = 32, 720, 1280, 32
t = synth.torch_inputs(dev, B, 3, H, W, flow_kind="smooth")
f0 = t["flow"]
cnt, out = f0.new_zeros((B, 1, H, W)), torch.zeros_like(f0)
ntiles = ((W + 63) // 64) * ((H + TH - 1) // TH) * B
buf = torch.zeros(ntiles * 16, dtype=torch.int64, device=dev)
setter = M.lib().memc_debug_set_trace_buffer_proj
setter.argtypes = [ctypes.c_void_p]
assert setter(ctypes.c_void_p(buf.data_ptr())) == 0
MARKS = [(1, "issue scan loads + zero P"), (2, "barrier + wait for the loads"), (3, "scan + splats (wave 0)"),
         (4, "barrier (slowest wave)"), (10, "box sums + normalise"), (6, "fill: any hole in the tile? (barrier)"),
         (7, "fill: masks + hole list + stage (barrier)"), (8, "fill: walks in the tile (barrier)"),
         (9, "fill: read back, summaries, masks"), (5, "store")]
FILL_SLOTS = (6, 7, 8, 9)
CASES = [(1.0, 0.0), (1.0, 40.0)] if len(sys.argv) < 2 else [(float(a.split(":")[0]), float(a.split(":")[1])) for a in sys.argv[1:]]
for scale, pan in CASES:                       # (argv: scale:pan pairs, e.g. 1:0 2:0 1:40)
    for fill in (0, 1):
        f = f0 * scale
        f[:, 0] += pan
        f[:, 1] -= pan / 2
        fn = lambda: L.FlowProjectionLayer_gpu_forward(f, cnt, out, fill)      # noqa: E731
        for _ in range(20):
            fn()
        buf.zero_()
        M.set_variant("projection", -41)
        torch.cuda.synchronize()
        fn()
        torch.cuda.synchronize()
        M.set_variant("projection", -1)
        ts = buf.cpu().numpy().reshape(ntiles, 16).astype(np.int64)
        ts = ts[ts[:, 5] > 0]
        print("flow x %g, pan %g px, fillhole %d: %d workgroups; clocks per phase: mean | p50 | p90 | p99 | max" % (scale, pan, fill, len(ts)))
        prev = 0
        for slot, nm in MARKS:
            if slot in FILL_SLOTS and not fill:
                continue
            d = (ts[:, slot] - ts[:, prev]).astype(np.float64)
            print("  %-46s %8.0f | %8.0f | %8.0f | %8.0f | %8.0f" % (nm, d.mean(), *np.percentile(d, [50, 90, 99]), d.max()))
            prev = slot
        life = (ts[:, 5] - ts[:, 0]).astype(np.float64)
        print("  %-46s %8.0f | %8.0f | %8.0f | %8.0f | %8.0f" % ("workgroup life", life.mean(), *np.percentile(life, [50, 90, 99]), life.max()))
