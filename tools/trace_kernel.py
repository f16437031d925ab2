#!/usr/bin/env python
"""tools/trace_kernel.py -- per-phase timeline of fi_bwd_tiled_c3 from in-kernel shader-clock timestamps
(measurement arm 9).  Run on the GPU box:  python tools/trace_kernel.py
Prints the mean / median duration of every phase of a workgroup and how many workgroups overlap on a CU."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "memc-net_amd"))
import my_package._ext.my_lib as L      # noqa: E402
from tools import synth                 # noqa: E402

# slot i -> what ended there
MARKS = [(1, "load inputs"), (2, "locate + bbox"), (3, "stage image"), (4, "phase 1 (taps/flow grads)"),
         (5, "zero plane"), (6, "adds c0"), (7, "flush c0"), (8, "adds c1"), (9, "flush c1"), (10, "adds c2"),
         (11, "flush c2"), (12, "later bands / tail")]


def main():
    dev = torch.device("cuda:0")
    B, C, H, W = 32, 3, 720, 1280
    t = synth.torch_inputs(dev, B, C, H, W, flow_kind="smooth", with_grad=True)
    x, f, k, g = t["x"], t["flow"], t["filt"], t["gout"]
    g1, g2, g3 = torch.zeros_like(x), torch.zeros_like(f), torch.zeros_like(k)
    ntiles = ((W + 63) // 64) * ((H + 15) // 16) * B
    nblk = ntiles
    buf = torch.zeros(nblk * 16, dtype=torch.int64, device=dev)
    lib = L._lib
    lib.memc_debug_set_trace_buffer.argtypes = [ctypes.c_void_p]
    assert lib.memc_debug_set_trace_buffer(ctypes.c_void_p(buf.data_ptr())) == 0
    for _ in range(30):
        L.FilterInterpolationLayer_gpu_backward(x, f, k, g, g1, g2, g3)
    L._debug_set_variant("fi_bwd", 9)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); L.FilterInterpolationLayer_gpu_backward(x, f, k, g, g1, g2, g3); b.record(); b.synchronize()
    L._debug_set_variant("fi_bwd", -1)
    ts = buf.cpu().numpy().reshape(nblk, 16).astype(np.int64)
    us = a.elapsed_time(b) * 1e3
    slots = 2 * torch.cuda.get_device_properties(0).multi_processor_count
    per_tile = us / (ntiles / slots)
    print("kernel %.1f us by events (timestamp arm); %d workgroups, %d resident -> %.2f us per tile" % (
        us, nblk, slots, per_tile))
    tot = (ts[:, 12] - ts[:, 0]).astype(np.float64)
    print("%-34s %9s %9s" % ("phase", "share", "~us"))
    prev = 0
    for slot, nm in MARKS:
        d = (ts[:, slot] - ts[:, prev]).astype(np.float64)
        print("%-34s %8.1f%% %9.2f" % (nm, 100 * d.mean() / tot.mean(), per_tile * d.mean() / tot.mean()))
        prev = slot


if __name__ == "__main__":
    main()
