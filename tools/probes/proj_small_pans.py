#!/usr/bin/env python
"""tools/probes/proj_small_pans.py -- FlowProjection + hole filling under SMALL camera pans (no far source: max |f| stays under
24 px): the uncovered band along two image edges is all holes whose upward / sideways walks cross tiles, i.e. work for
proj_fill_pending.  Flows: the benchmark's smooth flow x 0.25 (max |f| 4.4 px) + a pan (p, -p/2)."""
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
f0, d = t["flow"] * 0.25, t["depth"]
cnt, out = f0.new_zeros((32, 1, 720, 1280)), torch.zeros_like(f0)
for _ in range(100):
    L.FlowProjectionLayer_gpu_forward(f0, cnt, out, 0)
print("%-8s %8s %9s %12s %12s %14s" % ("pan", "max |f|", "holes %", "fill 0, us", "fill 1, us", "depth f1, us"))
for p in (0.0, 2.0, 4.0, 8.0, 12.0, 18.0):
    f = f0.clone()
    f[:, 0] += p
    f[:, 1] -= p / 2
    L.FlowProjectionLayer_gpu_forward(f, cnt, out, 0)
    holes = float((cnt == 0).float().mean()) * 100
    row = []
    for fn in (lambda: L.FlowProjectionLayer_gpu_forward(f, cnt, out, 0), lambda: L.FlowProjectionLayer_gpu_forward(f, cnt, out, 1),
               lambda: L.DepthFlowProjectionLayer_gpu_forward(f, d, cnt, out, 1)):
        for _ in range(3):
            fn()
        ts = []
        for _ in range(10):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); e1.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts.sort()
        row.append(ts[len(ts) // 2])
    print("%-8g %8.1f %9.2f %12.1f %12.1f %14.1f" % (p, float(f.abs().max()), holes, row[0], row[1], row[2]))
