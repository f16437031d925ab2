#!/usr/bin/env python
"""tools/probes/proj_burst.py -- the projection forward calls timed 1, 4 and 40 at a time between two events, with the
host's enqueue time per call: how much of a single call's figure is launch latency (three kernels, a stream-ordered
allocation and its release per call) rather than GPU time."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for _p in (ROOT, os.path.join(ROOT, "memc-net_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)
import torch  # noqa: E402

import my_package._ext.my_lib as L  # noqa: E402
from tools import synth  # noqa: E402

dev = torch.device("cuda:0")
t = synth.torch_inputs(dev, 32, 3, 720, 1280, flow_kind="smooth", with_depth=True)
f, d = t["flow"], t["depth"]
cnt, out = f.new_zeros((32, 1, 720, 1280)), torch.zeros_like(f)
for _ in range(150):
    L.FlowProjectionLayer_gpu_forward(f, cnt, out, 0)
CASES = (("proj fill0", lambda: L.FlowProjectionLayer_gpu_forward(f, cnt, out, 0)),
         ("proj fill1", lambda: L.FlowProjectionLayer_gpu_forward(f, cnt, out, 1)),
         ("depth fill1", lambda: L.DepthFlowProjectionLayer_gpu_forward(f, d, cnt, out, 1)))
for name, fn in CASES:
    for burst in (1, 4, 40):
        ts = []
        for rep in range(10):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            e0.record()
            for _ in range(burst):
                fn()
            e1.record()
            host = time.perf_counter() - t0
            e1.synchronize()
            ts.append((e0.elapsed_time(e1) * 1e3 / burst, host * 1e6 / burst))
        ts.sort()
        print("%-12s burst %2d: %7.1f us per call by events, host enqueue %6.1f us per call" % (
            name, burst, ts[len(ts) // 2][0], ts[len(ts) // 2][1]))
