#!/usr/bin/env python
"""tools/probes/host_overhead.py -- what a call costs the HOST (Python binder + C launcher, no synchronisation) at BASELINE
config 1's size (1x3x128x128), where the kernels take a few microseconds: raw bindings and the autograd modules."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for _p in (ROOT, os.path.join(ROOT, "memc-net_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)
import torch  # noqa: E402

import my_package._ext.my_lib as L  # noqa: E402
from my_package.modules.FilterInterpolationModule import FilterInterpolationModule  # noqa: E402
from my_package.modules.FlowProjectionModule import FlowProjectionModule  # noqa: E402
from tools import synth  # noqa: E402

dev = torch.device("cuda:0")
t = synth.torch_inputs(dev, 1, 3, 128, 128, flow_kind="smooth", with_grad=True)
x, f, k, go = t["x"], t["flow"], t["filt"], t["gout"]
out, g1, g2, g3 = torch.zeros_like(x), torch.zeros_like(x), torch.zeros_like(f), torch.zeros_like(k)
cnt, po = f.new_zeros((1, 1, 128, 128)), torch.zeros_like(f)


def host(fn, n=2000):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    dt = time.perf_counter() - t0
    torch.cuda.synchronize()
    return dt / n * 1e6


rows = [("FilterInterpolation fwd, raw binding", lambda: L.FilterInterpolationLayer_gpu_forward(x, f, k, out)),
        ("FilterInterpolation bwd, raw binding", lambda: L.FilterInterpolationLayer_gpu_backward(x, f, k, go, g1, g2, g3)),
        ("FlowProjection fwd (fill 1), raw binding", lambda: L.FlowProjectionLayer_gpu_forward(f, cnt, po, 1)),
        ("torch.empty_like (for scale)", lambda: torch.empty_like(x)),
        ("x + x (for scale: one ATen kernel)", lambda: x + x)]
fi, fp = FilterInterpolationModule(), FlowProjectionModule(requires_grad=False)
xr, fr, kr = x.clone().requires_grad_(True), f.clone().requires_grad_(True), k.clone().requires_grad_(True)


def train_step():
    o = fi(xr, fr, kr)
    o.backward(go)


with torch.no_grad():
    rows.append(("FilterInterpolationModule forward (no grad)", lambda: fi(x, f, k)))
    rows.append(("FlowProjectionModule forward (no grad)", lambda: fp(f)))
    for name, fn in rows:
        print("%-48s %7.1f us per call on the host" % (name, host(fn)))
print("%-48s %7.1f us per call on the host" % ("FilterInterpolationModule forward + backward", host(train_step, 500)))
