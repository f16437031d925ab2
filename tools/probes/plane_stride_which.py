#!/usr/bin/env python
"""tools/probes/plane_stride_which.py -- WHICH operand's layout is it?  The RGB forward at 720p batch 32 with the row padding of
tools/synth.py: padded_planes() applied to one operand at a time (the C ABI asks the output to share the image's b / c strides)."""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for _p in (ROOT, os.path.join(ROOT, "memc-net_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)
import torch  # noqa: E402

import my_package._ext.my_lib as L  # noqa: E402
from tools import synth  # noqa: E402

dev = torch.device("cuda:0")
t = synth.torch_inputs(dev, 32, 3, 720, 1280, flow_kind="smooth")
x, f, k = t["x"], t["flow"], t["filt"]
o = torch.zeros_like(x)
P = synth.padded_planes
px, pf, pk, po = P(x), P(f), P(k), P(o)
cases = {"all contiguous": (x, f, k, o), "taps padded": (x, f, pk, o), "flow padded": (x, pf, k, o), "image + output padded": (px, f, k, po),
         "taps + flow padded": (x, pf, pk, o), "all padded": (px, pf, pk, po)}
ts = {n: [] for n in cases}
for _ in range(100):
    L.FilterInterpolationLayer_gpu_forward(x, f, k, o)
for r in range(6):
    for n, a in cases.items():
        for _ in range(3):
            assert L.FilterInterpolationLayer_gpu_forward(*a) == 0, n
        for _ in range(8):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); L.FilterInterpolationLayer_gpu_forward(*a); e1.record(); e1.synchronize()
            ts[n].append(e0.elapsed_time(e1) * 1e3)
base = statistics.median(ts["all contiguous"])
for n in cases:
    m = statistics.median(ts[n])
    print("%-24s %7.1f us (%.3f)" % (n, m, m / base))
