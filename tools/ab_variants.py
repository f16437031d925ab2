#!/usr/bin/env python
"""tools/ab_variants.py -- same-process A/B of kernel VARIANTS of the measurement build (tools/measure.py): launches
alternate A, B, A, B ... in rounds on one GPU, the median of each is printed (box-to-box spread on the pool is +-4 %,
more than most single changes are worth).

    python tools/ab_variants.py --op projection --variants=-1,-40 [--cases proj,proj_fill,depth_fill] [--flows smooth,iid]
    python tools/ab_variants.py --op fi_bwd --variants=-1,50 --cases fi_bwd_c2,fi_bwd
"""
import argparse
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "memc-net_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)
import torch  # noqa: E402

from tools import measure as M  # noqa: E402
from tools import synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--op", default="projection")
    ap.add_argument("--variants", default="-1,-40")
    ap.add_argument("--cases", default="proj,proj_fill,depth,depth_fill")
    ap.add_argument("--flows", default="smooth")
    ap.add_argument("--rounds", type=int, default=6)
    ap.add_argument("--iters", type=int, default=12)
    ap.add_argument("--pan", type=float, default=0.0, help="camera pan (p, -p/2) px added to the flow")
    ap.add_argument("--scale", type=float, default=1.0, help="the flow times this")
    ap.add_argument("--lib", default="", help="another measurement build to bind (default: lib/libmemc_hip_measure.so)")
    a = ap.parse_args()
    if a.lib:
        M.MEASURE_LIB = os.path.abspath(a.lib)
    M.use()
    L = M.bound()
    dev = torch.device("cuda:0")
    variants = [int(v) for v in a.variants.split(",")]
    for flow in a.flows.split(","):
        B, H, W = 32, 720, 1280
        t = synth.torch_inputs(dev, B, 3, H, W, flow_kind=flow, with_depth=True, with_grad=True)
        x, f, k, g, d = t["x"], t["flow"], t["filt"], t["gout"], t["depth"]
        if a.scale != 1.0 or a.pan != 0.0:
            f = (f * a.scale).contiguous()
            f[:, 0] += a.pan
            f[:, 1] -= a.pan / 2
        cnt, out = f.new_zeros((B, 1, H, W)), torch.zeros_like(f)
        g1, g2, g3 = torch.zeros_like(x), torch.zeros_like(f), torch.zeros_like(k)
        t2 = synth.torch_inputs(dev, 8, 3, 256, 448, flow_kind=flow, with_grad=True)
        h1, h2, h3 = torch.zeros_like(t2["x"]), torch.zeros_like(t2["flow"]), torch.zeros_like(t2["filt"])
        ops = {
            "proj": (lambda: L.FlowProjectionLayer_gpu_forward(f, cnt, out, 0), None, 1),
            "proj_fill": (lambda: L.FlowProjectionLayer_gpu_forward(f, cnt, out, 1), None, 1),
            "depth": (lambda: L.DepthFlowProjectionLayer_gpu_forward(f, d, cnt, out, 0), None, 1),
            "depth_fill": (lambda: L.DepthFlowProjectionLayer_gpu_forward(f, d, cnt, out, 1), None, 1),
            "fi_bwd": (lambda: L.FilterInterpolationLayer_gpu_backward(x, f, k, g, g1, g2, g3), g1, 1),
            "fi_bwd_c2": (lambda: L.FilterInterpolationLayer_gpu_backward(t2["x"], t2["flow"], t2["filt"], t2["gout"], h1, h2, h3), h1, 20),
            "fi_bwd_nog1": (lambda: L.FilterInterpolationLayer_gpu_backward(x, f, k, g, None, g2, g3), None, 1),
            "fi_bwd_c2_nog1": (lambda: L.FilterInterpolationLayer_gpu_backward(t2["x"], t2["flow"], t2["filt"], t2["gout"], None, h2, h3), None, 20),
            "fi_fwd": (lambda: L.FilterInterpolationLayer_gpu_forward(x, f, k, g1), None, 1),      # the headline (--op fi_fwd)
            "bl_bwd": (lambda: L.InterpolationLayer_gpu_backward(x, f, g, g1, g2), g1, 1),
            # (with --op walk: the stripe width of the tile walk, -1 = the default)
            "bl_fwd": (lambda: L.InterpolationLayer_gpu_forward(x, f, g1), None, 1),
            "proj_bwd": (lambda: L.FlowProjectionLayer_gpu_backward(f, cnt0, gf, g2), None, 1),
            "depth_bwd": (lambda: L.DepthFlowProjectionLayer_gpu_backward(f, d, dcnt0, dout0, gf, g2, gd), None, 1),
        }
        gf, gd = torch.rand_like(f), torch.zeros_like(d)
        cnt0, out0 = torch.zeros_like(cnt), torch.zeros_like(out)
        L.FlowProjectionLayer_gpu_forward(f, cnt0, out0, 0)
        dcnt0, dout0 = torch.zeros_like(cnt), torch.zeros_like(out)
        L.DepthFlowProjectionLayer_gpu_forward(f, d, dcnt0, dout0, 0)
        for _ in range(150):
            ops["proj"][0]()
        for case in a.cases.split(","):
            fn, zero, burst = ops[case]
            ts = {v: [] for v in variants}
            for r in range(a.rounds):
                for v in variants:
                    M.set_variant(a.op, v)
                    for _ in range(4):
                        fn()
                    for _ in range(a.iters):
                        if zero is not None:
                            zero.zero_()
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        for _ in range(burst):
                            fn()
                        e1.record(); e1.synchronize()
                        ts[v].append(e0.elapsed_time(e1) * 1e3 / burst)
            M.set_variant(a.op, -1)
            med = {v: statistics.median(ts[v]) for v in variants}
            base = med[variants[0]]
            print("%-11s flow=%-6s " % (case, flow) + "   ".join("v%-4d %8.1f us (%.3f)" % (v, med[v], med[v] / base) for v in variants), flush=True)


if __name__ == "__main__":
    main()
