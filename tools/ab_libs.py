#!/usr/bin/env python
"""tools/ab_libs.py -- same-session A/B of two BUILDS of the library (box-to-box spread on the pool is +-3 %, more
than most single changes are worth: two builds can only be compared inside one process on one GPU).

    python tools/ab_libs.py <libA.so> <libB.so> [--op proj|proj_fill|depth_fill|fi_fwd|fi_bwd|fi_bwd_c64|interp_bwd_c64|ctx_img_blend|...] [--rounds 6] [--pan 40] [--scale 2]

Both libraries are loaded side by side (RTLD_LOCAL) and bound with my_package's own binder; launches alternate
A, B, A, B ... in rounds, the median of each is printed."""
import argparse
import ctypes
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "memc-net_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)
import torch  # noqa: E402

import my_package._ext.my_lib as L  # noqa: E402
from tools import synth  # noqa: E402


class Bound(object):
    def __init__(self, path):
        lib = ctypes.CDLL(path)
        lib.memc_hip_version.restype = ctypes.c_char_p
        self.version = lib.memc_hip_version().decode()
        for name, (n, flag) in list(L._SYMBOLS.items()) + list(L._EXTENSIONS.items()):
            if hasattr(lib, name):
                setattr(self, name, L._bind(name, n, flag, lib=lib, optional=L._OPTIONAL.get(name, ())))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs=2)
    ap.add_argument("--op", default="proj,proj_fill,depth_fill")
    ap.add_argument("--rounds", type=int, default=6)
    ap.add_argument("--iters", type=int, default=12)
    ap.add_argument("--pan", type=float, default=0.0, help="camera pan (p, -p/2) px added to the benchmark's flow")
    ap.add_argument("--scale", type=float, default=1.0, help="the benchmark's flow times this")
    ap.add_argument("--prezero", action="store_true", help="zero-fill the projection's outputs in front of every timed call, outside "
                    "the timed span (what a caller that follows the reference's contract does: FlowProjectionLayer.py:27-28)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    libs = [Bound(os.path.abspath(p)) for p in a.libs]
    B, H, W = 32, 720, 1280
    t = synth.torch_inputs(dev, B, 3, H, W, flow_kind="smooth", with_depth=True, with_grad=True)
    x, f, k, g, d = t["x"], t["flow"], t["filt"], t["gout"], t["depth"]
    if a.scale != 1.0 or a.pan != 0.0:
        f = (f * a.scale).contiguous()
        f[:, 0] += a.pan
        f[:, 1] -= a.pan / 2
    cnt, out = f.new_zeros((B, 1, H, W)), torch.zeros_like(f)
    o3 = torch.zeros_like(x)
    g1, g2, g3 = torch.zeros_like(x), torch.zeros_like(f), torch.zeros_like(k)
    g2src, gd = torch.rand_like(f), torch.zeros_like(d)
    c2 = synth.torch_inputs(dev, 8, 3, 256, 448, flow_kind="smooth")
    c2o = torch.zeros_like(c2["x"])
    ops = {
        "proj": lambda l: l.FlowProjectionLayer_gpu_forward(f, cnt, out, 0),
        "proj_fill": lambda l: l.FlowProjectionLayer_gpu_forward(f, cnt, out, 1),
        "depth_fill": lambda l: l.DepthFlowProjectionLayer_gpu_forward(f, d, cnt, out, 1),
        "fi_fwd": lambda l: l.FilterInterpolationLayer_gpu_forward(x, f, k, o3),
        "fi_bwd": lambda l: l.FilterInterpolationLayer_gpu_backward(x, f, k, g, g1, g2, g3),
        "interp_fwd": lambda l: l.InterpolationLayer_gpu_forward(x, f, o3),
        "interp_bwd": lambda l: l.InterpolationLayer_gpu_backward(x, f, g, g1, g2),
        "proj_bwd": lambda l: l.FlowProjectionLayer_gpu_backward(f, cnt, g2src, g2),
        "depth_bwd": lambda l: l.DepthFlowProjectionLayer_gpu_backward(f, d, cnt, out, g2src, g2, gd),
        "fi_fwd_c2": lambda l: l.FilterInterpolationLayer_gpu_forward(c2["x"], c2["flow"], c2["filt"], c2o),
    }
    if any(o in a.op.split(",") for o in ("fi_bwd_c64", "interp_bwd_c64", "ctx_img_blend")):
        # the many-channel operators (8 x 64 x 720 x 1280: the context features of config 4's network), allocated on demand
        m = synth.torch_inputs(dev, 8, 64, H, W, flow_kind="smooth", with_grad=True)
        mg1, mg2, mg3 = torch.zeros_like(m["x"]), torch.zeros_like(m["flow"]), torch.zeros_like(m["filt"])
        mi = {k_: v[:8].contiguous() for k_, v in (("x", x), ("occ", d))}
        mprev, mio, mco = torch.rand_like(mi["x"]), torch.zeros_like(mi["x"]), torch.zeros_like(m["x"])
        ops["fi_bwd_c64"] = lambda l: l.FilterInterpolationLayer_gpu_backward(m["x"], m["flow"], m["filt"], m["gout"], mg1, mg2, mg3)
        ops["interp_bwd_c64"] = lambda l: l.InterpolationChLayer_gpu_backward(m["x"], m["flow"], m["gout"], mg1, mg2)
        ops["ctx_img_blend"] = lambda l: l.FilterInterpolationCtxLayer_gpu_forward(
            mi["x"], m["x"], m["flow"], m["filt"], mprev, mi["occ"], mi["occ"], mio, mco)
    libs[0].FlowProjectionLayer_gpu_forward(f, cnt, out, 0)
    for _ in range(150):
        ops["proj"](libs[0])
    for op in a.op.split(","):
        fn = ops[op]
        ts = [[], []]
        for r in range(a.rounds):
            for i, l in enumerate(libs):
                for _ in range(4):
                    fn(l)
                for _ in range(a.iters):
                    if op in ("fi_bwd", "interp_bwd"):
                        g1.zero_()
                    if a.prezero and op in ("proj", "proj_fill", "depth_fill"):
                        cnt.zero_(); out.zero_()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(); fn(l); e1.record(); e1.synchronize()
                    ts[i].append(e0.elapsed_time(e1) * 1e3)
        ma, mb = statistics.median(ts[0]), statistics.median(ts[1])
        print("%-12s A %8.1f us   B %8.1f us   B/A %.3f   (A = %s, B = %s)" % (
            op, ma, mb, mb / ma, os.path.basename(a.libs[0]), os.path.basename(a.libs[1])), flush=True)


if __name__ == "__main__":
    main()
