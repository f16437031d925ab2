#!/usr/bin/env python
"""tools/probes/proj_motion_sweep.py -- FlowProjection forward (with hole filling) against the size of the motion: the
benchmark's smooth flow scaled by 1, 1.5, 2, 3, 4, 6 (|flow| up to ~18 px at scale 1).  Sources that move 24 px or more
are "far": their images are redone by proj_owner_far -- what does that cost, and from which motion on?"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for _p in (ROOT, os.path.join(ROOT, "memc-net_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)
import torch  # noqa: E402

import my_package._ext.my_lib as L  # noqa: E402
from tools import synth  # noqa: E402

dev = torch.device("cuda:0")
t = synth.torch_inputs(dev, 32, 3, 720, 1280, flow_kind="smooth", with_depth=True)
f0, d = t["flow"], t["depth"]
cnt, out = f0.new_zeros((32, 1, 720, 1280)), torch.zeros_like(f0)
for _ in range(100):
    L.FlowProjectionLayer_gpu_forward(f0, cnt, out, 0)
print("%-7s %10s %12s %14s %14s %16s" % ("scale", "max |f|", "far sites %", "fill 0, us", "fill 1, us", "depth fill 1, us"))
for scale in (1.0, 1.5, 2.0, 3.0, 4.0, 6.0):
    f = (f0 * scale).contiguous()
    far = float(((f.abs() >= 24).any(dim=1)).float().mean()) * 100
    row = []
    for fn in (lambda: L.FlowProjectionLayer_gpu_forward(f, cnt, out, 0), lambda: L.FlowProjectionLayer_gpu_forward(f, cnt, out, 1),
               lambda: L.DepthFlowProjectionLayer_gpu_forward(f, d, cnt, out, 1)):
        for _ in range(3):
            fn()
        ts = []
        for _ in range(8):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); e1.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts.sort()
        row.append(ts[len(ts) // 2])
    print("%-7.1f %10.1f %12.3f %14.1f %14.1f %16.1f" % (scale, float(f.abs().max()), far, row[0], row[1], row[2]))
