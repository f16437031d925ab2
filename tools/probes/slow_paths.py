#!/usr/bin/env python
"""tools/probes/slow_paths.py -- what the scalar / direct kernels cost: every operator at 720p batch 32 on (a) the aligned
shape, (b) a width that is not a multiple of four (1278), (c) the aligned width seen through a view that starts one
element in -- with the kernel family the library reports for each call (memc_last_kernel_path).
Output: profiles/r04_slow_paths.txt (copied from gpurun_out by the session)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for _p in (ROOT, os.path.join(ROOT, "memc-net_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)
import torch  # noqa: E402

import my_package._ext.my_lib as L  # noqa: E402

dev = torch.device("cuda:0")
B, H = 32, 720


def timed(fn, pre=None, iters=12):
    for _ in range(3):
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


def case(name, W, sliced):
    Wa = W + 1 if sliced else W

    def view(t):
        return t[..., 1:] if sliced else t

    g = torch.Generator(device=dev); g.manual_seed(3)
    x = view(torch.rand((B, 3, H, Wa), device=dev, generator=g))
    # a smooth flow of a few pixels (x4 bilinear upsampling of noise)
    f = view(torch.nn.functional.interpolate(torch.randn((B, 2, H // 16 + 1, Wa // 16 + 1), device=dev, generator=g) * 4.0,
                                             size=(H, Wa), mode="bilinear", align_corners=True).contiguous())
    k = view(torch.rand((B, 16, H, Wa), device=dev, generator=g) / 16)
    go = view(torch.rand((B, 3, H, Wa), device=dev, generator=g))
    out, g1 = view(torch.zeros((B, 3, H, Wa), device=dev)), view(torch.zeros((B, 3, H, Wa), device=dev))
    g2, g3 = view(torch.zeros((B, 2, H, Wa), device=dev)), view(torch.zeros((B, 16, H, Wa), device=dev))
    cnt, po = view(torch.zeros((B, 1, H, Wa), device=dev)), view(torch.zeros((B, 2, H, Wa), device=dev))
    rows = []

    def row(op, fn, pre=None):
        us = timed(fn, pre)
        rows.append((op, L.last_kernel_path(), us))
    row("FilterInterpolation fwd", lambda: L.FilterInterpolationLayer_gpu_forward(x, f, k, out))
    row("FilterInterpolation bwd", lambda: L.FilterInterpolationLayer_gpu_backward(x, f, k, go, g1, g2, g3), lambda: g1.zero_())
    row("Interpolation fwd", lambda: L.InterpolationLayer_gpu_forward(x, f, out))
    row("Interpolation bwd", lambda: L.InterpolationLayer_gpu_backward(x, f, go, g1, g2), lambda: g1.zero_())
    row("FlowProjection fwd fill=1", lambda: L.FlowProjectionLayer_gpu_forward(f, cnt, po, 1))
    row("FlowProjection bwd", lambda: L.FlowProjectionLayer_gpu_backward(f, cnt, po, g2))
    print("%s (%dx3x%dx%d%s)" % (name, B, H, W, ", rows start one element into a 16-byte unit" if sliced else ""))
    for op, path, us in rows:
        print("   %-28s %-22s %9.1f us" % (op, path, us))
    return rows


base = case("aligned", 1280, False)
for name, W, sliced in (("odd width", 1278, False), ("unaligned view", 1280, True)):
    rows = case(name, W, sliced)
    print("   slow-down against the aligned shape: " + ", ".join("%s %.1fx" % (r[0], r[2] / b[2]) for r, b in zip(rows, base)))
