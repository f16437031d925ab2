#!/usr/bin/env python
"""tools/probes/motion_sweep_all.py -- every operator at 32x3x720x1280 against the size of the motion: the benchmark's smooth
flow scaled by 1 ... 6 (max |f| 17.5 ... 105 px; its gradient scales too, so the boxes a tile stages grow) and a rigid pan
of the same size added to the unscaled flow (boxes keep their size, only move).  With the kernel family the library
reports per call."""
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
B, H, W = 32, 720, 1280
t = synth.torch_inputs(dev, B, 3, H, W, flow_kind="smooth", with_depth=True, with_grad=True)
x, f0, k, d = t["x"], t["flow"], t["filt"], t["depth"]
go = t["gout"]
go2 = go[:, :2].contiguous()
out, g1 = torch.zeros_like(x), torch.zeros_like(x)
g2, g3 = torch.zeros_like(f0), torch.zeros_like(k)
cnt, po, gd = f0.new_zeros((B, 1, H, W)), torch.zeros_like(f0), f0.new_zeros((B, 1, H, W))


def timed(fn, pre=None, iters=8):
    for _ in range(2):
        if pre:
            pre()
        fn()
    ts = []
    for _ in range(iters):
        if pre:
            pre()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def sweep(name, flows):
    print(name)
    print("  %-10s %8s | %9s %9s | %9s %9s | %9s %9s %9s" % ("", "max |f|", "FI fwd", "FI bwd", "Interp f", "Interp b", "Proj f1", "Proj bwd",
                                                              "Depth bwd"))
    for label, f in flows:
        f = f.contiguous()
        L.FlowProjectionLayer_gpu_forward(f, cnt, po, 0)
        row = [timed(lambda: L.FilterInterpolationLayer_gpu_forward(x, f, k, out)),
               timed(lambda: L.FilterInterpolationLayer_gpu_backward(x, f, k, go, g1, g2, g3), lambda: g1.zero_()),
               timed(lambda: L.InterpolationLayer_gpu_forward(x, f, out)),
               timed(lambda: L.InterpolationLayer_gpu_backward(x, f, go, g1, g2), lambda: g1.zero_()),
               timed(lambda: L.FlowProjectionLayer_gpu_forward(f, cnt, po, 1))]
        L.FlowProjectionLayer_gpu_forward(f, cnt, po, 0)
        row.append(timed(lambda: L.FlowProjectionLayer_gpu_backward(f, cnt, go2, g2)))
        L.DepthFlowProjectionLayer_gpu_forward(f, d, cnt, po, 0)
        row.append(timed(lambda: L.DepthFlowProjectionLayer_gpu_backward(f, d, cnt, po, go2, g2, gd)))
        print("  %-10s %8.1f | %9.1f %9.1f | %9.1f %9.1f | %9.1f %9.1f %9.1f" % ((label, float(f.abs().max())) + tuple(row)))


sweep("scaled: the benchmark's flow x s", [("x %.1f" % s, f0 * s) for s in (1.0, 1.5, 2.0, 3.0, 4.0, 6.0)])
pan = torch.zeros_like(f0)
flows = []
for p in (0.0, 20.0, 40.0, 80.0, 160.0):
    q = f0.clone()
    q[:, 0] += p
    q[:, 1] -= p / 2
    flows.append(("pan %g" % p, q))
sweep("panned: the benchmark's flow + (p, -p/2)", flows)

# a fast OBJECT over a slow background: the benchmark's flow x 0.5 with a rectangle of 300 x 200 px per image moving (v, -v/2)
flows = []
for v in (0.0, 20.0, 40.0, 80.0):
    q = f0 * 0.5
    for b in range(B):
        y0, x0 = 60 + 13 * b, 100 + 27 * b
        q[b, 0, y0:y0 + 200, x0:x0 + 300] = v
        q[b, 1, y0:y0 + 200, x0:x0 + 300] = -v / 2
    flows.append(("object %g" % v, q))
sweep("a fast object: the benchmark's flow x 0.5, a rectangle of 300 x 200 px moving (v, -v/2)", flows)
