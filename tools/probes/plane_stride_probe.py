#!/usr/bin/env python
"""tools/probes/plane_stride_probe.py -- do the PLANE and ROW STRIDES of the operands matter?  A 1280 x 720 fp32 plane is 3,686,400 B =
14400 x 256 B and a row 5120 B = 20 x 256 B: the sixteen tap planes a lane reads at one position, and the sixteen rows of a tile, lie
multiples of 4 x 256 B apart.  The library takes any strides (views), so the same calls are timed on tensors whose channel stride (and / or
row stride) is padded.  One process; results checked against the contiguous call.
    python tools/probes/plane_stride_probe.py [fi_fwd|all]"""
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
which = sys.argv[1] if len(sys.argv) > 1 else "fi_fwd"


def padded(src, cpad, rpad):
    """the same values in a buffer whose rows are rpad floats longer and whose channel planes are cpad floats further apart"""
    b, c, h, w = src.shape
    buf = torch.empty((b, c, h * (w + rpad) + cpad), device=dev)
    view = buf[:, :, :h * (w + rpad)].view(b, c, h, w + rpad)[:, :, :, :w]
    view.copy_(src)
    return view


def timeit(fn, rounds_unused=None):
    for _ in range(3):
        fn()
    ts = []
    for _ in range(8):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    return ts


def sweep(name, shape, make_call, layouts, rounds=5):
    """make_call(layout) -> (callable, output tensor, reference output or None)"""
    calls = {lay: make_call(lay) for lay in layouts}
    ts = {lay: [] for lay in layouts}
    for r in range(rounds):
        for lay in layouts:
            ts[lay] += timeit(calls[lay][0])
    base = statistics.median(ts[layouts[0]])
    ref = calls[layouts[0]][1]
    for lay in layouts:
        m = statistics.median(ts[lay])
        same = bool(torch.equal(calls[lay][1], ref)) if calls[lay][2] else None
        print("%-28s %-16s channel stride + %6d B, row stride + %5d B: %8.1f us (%.3f)%s" % (
            name, "x".join(map(str, shape)), 4 * lay[0], 4 * lay[1], m, m / base, "" if same in (True, None) else "  RESULT DIFFERS"), flush=True)


LAYOUTS = [(0, 0), (64, 0), (192, 0), (320, 0), (448, 0), (1088, 0), (4160, 0), (0, 64), (64, 64), (0, 192), (1024, 0), (4096, 0)]

B, H, W = 32, 720, 1280
t = synth.torch_inputs(dev, B, 3, H, W, flow_kind="smooth", with_grad=True, with_depth=True)


def fi_fwd(lay):
    x, f, k = (padded(t[n], *lay) for n in ("x", "flow", "filt"))
    o = padded(torch.zeros_like(t["x"]), *lay)
    return (lambda: L.FilterInterpolationLayer_gpu_forward(x, f, k, o)), o, True


for _ in range(100):
    L.FilterInterpolationLayer_gpu_forward(t["x"], t["flow"], t["filt"], torch.empty_like(t["x"]))
sweep("FilterInterpolation fwd C=3", (B, 3, H, W), fi_fwd, LAYOUTS)
if which == "all":
    SHORT = [(0, 0), (64, 0), (4160, 0), (0, 64), (64, 64)]

    def fi_bwd(lay):
        x, f, k, g = (padded(t[n], *lay) for n in ("x", "flow", "filt", "gout"))
        g1, g2, g3 = (padded(torch.zeros_like(t[n]), *lay) for n in ("x", "flow", "filt"))

        def call():
            g1.zero_()
            L.FilterInterpolationLayer_gpu_backward(x, f, k, g, g1, g2, g3)
        return call, g3, False
    sweep("FilterInterpolation bwd C=3", (B, 3, H, W), fi_bwd, SHORT)

    def proj(lay):
        f = padded(t["flow"], *lay)
        cnt, po = padded(torch.zeros((B, 1, H, W), device=dev), *lay), padded(torch.zeros_like(t["flow"]), *lay)
        return (lambda: L.FlowProjectionLayer_gpu_forward(f, cnt, po, 1)), cnt, True
    sweep("FlowProjection fwd + fill", (B, 2, H, W), proj, SHORT)

    def bl_fwd(lay):
        x, f = padded(t["x"], *lay), padded(t["flow"], *lay)
        o = padded(torch.zeros_like(t["x"]), *lay)
        return (lambda: L.InterpolationLayer_gpu_forward(x, f, o)), o, True
    sweep("Interpolation fwd C=3", (B, 3, H, W), bl_fwd, SHORT)
    del t
    torch.cuda.empty_cache()
    t = synth.torch_inputs(dev, 8, 64, H, W, flow_kind="smooth")
    sweep("FilterInterpolation fwd C=64", (8, 64, H, W), fi_fwd, SHORT, rounds=3)
    del t
    torch.cuda.empty_cache()
    t = synth.torch_inputs(dev, 8, 3, 2160, 3840, flow_kind="smooth")
    sweep("FilterInterpolation fwd 4K", (8, 3, 2160, 3840), fi_fwd, SHORT, rounds=3)
    del t
    torch.cuda.empty_cache()
    t = synth.torch_inputs(dev, 8, 3, 256, 448, flow_kind="smooth")
    sweep("FilterInterpolation fwd cfg2", (8, 3, 256, 448), fi_fwd, SHORT, rounds=5)
