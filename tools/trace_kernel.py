#!/usr/bin/env python
"""tools/trace_kernel.py -- per-phase timeline of a tiled kernel from in-kernel shader-clock timestamps
(measurement arms: fi_bwd variant 9, projection variant -7).  Run on the GPU box:

    python tools/trace_kernel.py [fi_bwd|proj]

Prints the share of a workgroup's life spent in every phase (thread 0's view; barriers fold the slowest wave's
time into the phase that ends with them)."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "memc-net_amd"))
import my_package._ext.my_lib as L      # noqa: E402
from tools import measure as M  # noqa: E402
M.use()                             # the measurement build: ablation / A-B arms live only there
from tools import synth                 # noqa: E402

KERNELS = {
    "fi_bwd": dict(setter="memc_debug_set_trace_buffer", op="fi_bwd", variant=9, last=12,
                   marks=[(1, "load inputs"), (2, "locate + bbox"), (3, "stage image"),
                          (4, "phase 1 (taps/flow grads)"), (5, "zero plane"), (6, "adds c0"), (7, "flush c0"),
                          (8, "adds c1"), (9, "flush c1"), (10, "adds c2"), (11, "flush c2"),
                          (12, "later bands / tail")]),
    # round 3 (the product kernel + timestamps): packed fixed-point planes, image gradient first (fi_bwd_c3.hip)
    "fi_bwd_pk2": dict(setter="memc_debug_set_trace_buffer", op="fi_bwd", variant=28, last=12,
                       marks=[(1, "load inputs (planes zeroed meanwhile)"), (2, "locate + bbox + block exponent"),
                              (3, "request image rows; packed adds + barrier"), (4, "flush (3 colours) + barrier"),
                              (5, "image rows -> LDS + barrier"), (6, "phase 1 (taps/flow grads)"),
                              (12, "later bands / tail")]),
    "proj": dict(setter="memc_debug_set_trace_buffer_proj", op="projection", variant=-7, last=5, th=16,
                 marks=[(1, "issue scan loads + zero P + barrier"), (2, "wait for the loads"),
                        (3, "scan: locate + fp64 splat (wave 0)"), (4, "barrier (slowest wave)"),
                        (5, "box sum + normalise + store")]),
}
_PROJ2_MARKS = [(1, "issue scan loads + zero P + barrier"), (2, "wait for the loads"),
                (3, "scan: window test + compaction + dense fp64 splat (wave 0)"), (4, "barrier (slowest wave)"),
                (5, "box sum + normalise + store")]
# round 4: the production owner kernel (direct masked splats) with timestamps
KERNELS["proj5"] = dict(setter="memc_debug_set_trace_buffer_proj", op="projection", variant=-41, last=5, th=32,
                        marks=[(1, "issue scan loads + zero P"), (2, "barrier + wait for the loads"),
                               (3, "scan: window tests + masked fp64 splats (wave 0)"), (4, "barrier (slowest wave)"),
                               (5, "box sum + normalise + store")])
for _code, _th in ((0, 16), (1, 32), (2, 64)):      # proj_owner2 (round 2), tile height 16 / 32 / 64
    KERNELS["proj2_%d" % _th] = dict(setter="memc_debug_set_trace_buffer_proj", op="projection", variant=290 + _code,
                                     last=5, th=_th, marks=_PROJ2_MARKS)


def fi_bwd_cn():
    """fi_bwd_image_owner (fi_bwd_cn.hip): accumulated shader clocks per phase, thread 0 of every workgroup."""
    dev = torch.device("cuda:0")
    B, C, H, W = 8, 64, 720, 1280
    kind = sys.argv[2] if len(sys.argv) > 2 else "smooth"
    t = synth.torch_inputs(dev, B, C, H, W, flow_kind=kind, with_grad=True)
    x, f, k, g = t["x"], t["flow"], t["filt"], t["gout"]
    g1, g2, g3 = torch.zeros_like(x), torch.zeros_like(f), torch.zeros_like(k)
    fn = lambda: L.FilterInterpolationLayer_gpu_backward(x, f, k, g, g1, g2, g3)     # noqa: E731
    ntiles = ((W + 63) // 64) * ((H + 15) // 16) * B
    buf = torch.zeros(ntiles * 16, dtype=torch.int64, device=dev)
    setter = M.lib().memc_debug_set_trace_buffer_cn
    setter.argtypes = [ctypes.c_void_p]
    for _ in range(5):
        fn()
    assert setter(ctypes.c_void_p(buf.data_ptr())) == 0
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); fn(); b.record(); b.synchronize()
    setter(ctypes.c_void_p(0))
    ts = buf.cpu().numpy().reshape(ntiles, 16).astype(np.float64)
    print("fi_bwd C=%d %dx%dx%d flow=%s: %.1f us by events (both kernels, trace arm), %d workgroups" % (
        C, B, H, W, kind, a.elapsed_time(b) * 1e3, ntiles))
    tot = ts[:, 0].mean()
    print("mean workgroup life %.0f clocks (100 MHz counter ticks if s_memrealtime; see ratio only)" % tot)
    names = {1: "candidates + count pass", 2: "slab recounts", 3: "scan", 4: "fill", 5: "lists -> registers", 6: "replay (outside the marks below)",
             11: "  replay: stage + barrier", 12: "  replay: loads issue + segments", 13: "  replay: heads",
             14: "  replay: barrier (segment sums)", 15: "  replay: sums, transpose, store, barrier"}
    for i, nm in names.items():
        print("%-28s %6.1f%%  %10.0f" % (nm, 100 * ts[:, i].mean() / tot, ts[:, i].mean()))
    print("slab rounds per tile: mean %.2f max %d; candidate strips (64 x 4 sites) mean %.1f; longest list (wave 0, last slab) mean %.1f max %d; "
          "site box mean %.0f max %d" % (ts[:, 7].mean(), ts[:, 7].max(), ts[:, 8].mean(), ts[:, 9].mean(), ts[:, 9].max(),
                                         ts[:, 10].mean(), ts[:, 10].max()))


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "fi_bwd"
    if which == "fi_bwd_cn":
        return fi_bwd_cn()
    K = KERNELS[which]
    dev = torch.device("cuda:0")
    B, C, H, W = 32, 3, 720, 1280
    t = synth.torch_inputs(dev, B, C, H, W, flow_kind="smooth", with_grad=True)
    x, f, k, g = t["x"], t["flow"], t["filt"], t["gout"]
    if which.startswith("fi_bwd"):
        g1, g2, g3 = torch.zeros_like(x), torch.zeros_like(f), torch.zeros_like(k)
        fn = lambda: L.FilterInterpolationLayer_gpu_backward(x, f, k, g, g1, g2, g3)     # noqa: E731
    else:
        cnt, out = f.new_zeros((B, 1, H, W)), torch.zeros_like(f)
        fn = lambda: L.FlowProjectionLayer_gpu_forward(f, cnt, out, 0)                   # noqa: E731
    th = K.get("th", 16)
    ntiles = ((W + 63) // 64) * ((H + th - 1) // th) * B
    buf = torch.zeros(ntiles * 16, dtype=torch.int64, device=dev)
    setter = getattr(M.lib(), K["setter"])
    setter.argtypes = [ctypes.c_void_p]
    assert setter(ctypes.c_void_p(buf.data_ptr())) == 0
    for _ in range(30):
        fn()
    M.set_variant(K["op"], K["variant"])
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); fn(); b.record(); b.synchronize()
    M.set_variant(K["op"], -1)
    ts = buf.cpu().numpy().reshape(ntiles, 16).astype(np.int64)
    us = a.elapsed_time(b) * 1e3
    tot = (ts[:, K["last"]] - ts[:, 0]).astype(np.float64)
    print("%s: %.1f us by events (timestamp arm), %d workgroups, mean workgroup life %.0f shader clocks" % (
        which, us, ntiles, tot.mean()))
    print("%-40s %8s %12s" % ("phase", "share", "mean clocks"))
    prev = 0
    for slot, nm in K["marks"]:
        d = (ts[:, slot] - ts[:, prev]).astype(np.float64)
        print("%-40s %7.1f%% %12.0f" % (nm, 100 * d.mean() / tot.mean(), d.mean()))
        prev = slot


if __name__ == "__main__":
    main()
