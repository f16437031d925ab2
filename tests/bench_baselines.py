#!/usr/bin/env python
"""tests/bench_baselines.py -- the two baselines the operator sweep (tools/bench_ops.py) is read against, timed on
the same box.  Lives under tests/ because both are checkers, not product code:

  * the REFERENCE'S OWN GPU KERNELS (oracle/ref_gpu.py: my_lib_kernel.cu built for gfx950) on the benchmark sizes --
    what the unmodified reference would do on an MI355X, next to this repository's kernels for the same call;
  * the CPU oracle (port of the reference's CPU code, OpenMP over all host cores) on a bounded 720p sample.

    python tests/bench_baselines.py [--json gpurun_out/baselines.json] [--no-cpu]
"""
import argparse
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "memc-net_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402

from tools import synth  # noqa: E402


def _time(fn, warmup=3, iters=7):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize()
        ts.append(a.elapsed_time(b) * 1e-3)
    return statistics.median(ts)


def reference_gpu_rows(rows):
    """Reference kernels vs this repository's, same tensors, 720p batch 32 (batch 8 for C = 64).  Both sides are
    timed as a CALL: the reference wrappers allocate + zero-fill every output / gradient like the reference's
    Python layer, this repository's side allocates what its shipped Python layer allocates (zero-filled only where
    its kernels accumulate, uninitialised where they define every element)."""
    from oracle import ref_gpu as R
    if not R.available():
        print("reference kernels not built (make -C oracle ref needs /root/reference): skipped")
        return
    import my_package._ext.my_lib as L
    from my_package.functions.FilterInterpolationLayer import FilterInterpolationLayer
    from my_package.functions.FlowProjectionLayer import FlowProjectionLayer
    from my_package.functions.DepthFlowProjectionLayer import DepthFlowProjectionLayer
    dev = torch.device("cuda:0")
    B, C, H, W = 32, 3, 720, 1280
    t = synth.torch_inputs(dev, B, C, H, W, flow_kind="smooth", with_grad=True, with_depth=True)
    x, f, k, g, d = t["x"], t["flow"], t["filt"], t["gout"], t["depth"]
    gf = torch.rand_like(f)
    t64 = synth.torch_inputs(dev, 8, 64, H, W, flow_kind="smooth", with_grad=True)

    def ours_fi_bwd64():
        a, b, c, e = t64["x"], t64["flow"], t64["filt"], t64["gout"]
        g1, g2, g3 = torch.empty_like(a), torch.empty_like(b), torch.empty_like(c)      # as the shipped layer does
        L.FilterInterpolationLayer_gpu_backward(a, b, c, e, g1, g2, g3)

    def ours_fi_bwd():
        g1, g2, g3 = torch.zeros_like(x), torch.empty_like(f), torch.empty_like(k)      # as the shipped layer does
        L.FilterInterpolationLayer_gpu_backward(x, f, k, g, g1, g2, g3)

    def ours_fp(fill):
        cnt, out = f.new_empty((B, 1, H, W)), torch.empty_like(f)
        L.FlowProjectionLayer_gpu_forward(f, cnt, out, fill)

    def ours_dfp(fill):
        cnt, out = torch.empty_like(d), torch.empty_like(f)
        L.DepthFlowProjectionLayer_gpu_forward(f, d, cnt, out, fill)

    _, cnt0 = R.flow_projection_forward(f, 0)
    do0, dc0 = R.depth_flow_projection_forward(f, d, 0)

    def ours_fp_bwd():
        g1 = torch.empty_like(f)                                                        # as the shipped layer does
        L.FlowProjectionLayer_gpu_backward(f, cnt0, gf, g1)

    def ours_dfp_bwd():
        g1, g2 = torch.empty_like(f), torch.empty_like(d)
        L.DepthFlowProjectionLayer_gpu_backward(f, d, dc0, do0, gf, g1, g2)

    def ours_bl():
        out = torch.empty_like(x)
        L.InterpolationLayer_gpu_forward(x, f, out)

    def ours_bl_bwd():
        g1, g2 = torch.zeros_like(x), torch.empty_like(f)
        L.InterpolationLayer_gpu_backward(x, f, g, g1, g2)

    with torch.no_grad():
        pairs = [
            ("FilterInterpolation fwd C=3 32x720x1280", lambda: R.filter_interpolation_forward(x, f, k),
             lambda: FilterInterpolationLayer()(x, f, k)),
            ("FilterInterpolation fwd C=64 8x720x1280", lambda: R.filter_interpolation_forward(t64["x"], t64["flow"], t64["filt"]),
             lambda: FilterInterpolationLayer()(t64["x"], t64["flow"], t64["filt"])),
            ("FilterInterpolation bwd C=3 32x720x1280", lambda: R.filter_interpolation_backward(x, f, k, g), ours_fi_bwd),
            ("FilterInterpolation bwd C=64 8x720x1280",
             lambda: R.filter_interpolation_backward(t64["x"], t64["flow"], t64["filt"], t64["gout"]), ours_fi_bwd64),
            ("FlowProjection fwd 32x720x1280", lambda: R.flow_projection_forward(f, 0), lambda: ours_fp(0)),
            ("FlowProjection fwd + hole fill", lambda: R.flow_projection_forward(f, 1), lambda: ours_fp(1)),
            ("DepthFlowProjection fwd + hole fill", lambda: R.depth_flow_projection_forward(f, d, 1), lambda: ours_dfp(1)),
            ("FlowProjection bwd", lambda: R.flow_projection_backward(f, cnt0, gf), ours_fp_bwd),
            ("DepthFlowProjection bwd", lambda: R.depth_flow_projection_backward(f, d, dc0, do0, gf), ours_dfp_bwd),
            ("Interpolation fwd C=3", lambda: R.interpolation_forward(x, f), ours_bl),
            ("Interpolation bwd C=3", lambda: R.interpolation_backward(x, f, g), ours_bl_bwd),
        ]
        print("%-44s %14s %14s %9s" % ("call (outputs allocated + zeroed inside)", "reference us", "this repo us", "speed-up"))
        for name, ref_fn, our_fn in pairs:
            tr, to = _time(ref_fn), _time(our_fn)
            rows.append({"op": name, "reference_kernels_us": round(tr * 1e6, 1), "this_repo_us": round(to * 1e6, 1),
                         "speedup": round(tr / to, 2)})
            print("%-44s %14.1f %14.1f %8.1fx" % (name, tr * 1e6, to * 1e6, tr / to), flush=True)


def cpu_rows(rows):
    """The oracle (port of the reference's CPU code) timed per operator on a bounded 720p sample, same box:
    a baseline beside the GPU rows -- says nothing about kernel quality (that is the roofline fraction)."""
    import time
    import numpy as np
    from oracle import memc_oracle as O          # baseline only
    O.build()
    rng = np.random.default_rng(0)
    B, C, H, W = 4, 3, 720, 1280
    x = synth.np_image(rng, B, C, H, W); f = synth.np_flow(rng, B, H, W, "smooth"); k = synth.np_filter(rng, B, H, W)
    g = synth.np_image(rng, B, C, H, W); d = synth.np_depth(rng, B, H, W); gf = rng.random((B, 2, H, W), dtype=np.float32)
    _, cnt = O.flow_projection_forward(f, 0)
    do, dc = O.depth_flow_projection_forward(f, d, 0)
    cases = [
        ("fi_fwd", lambda: O.filter_interpolation_forward(x, f, k)),
        ("fi_bwd", lambda: O.filter_interpolation_backward(x, f, k, g)),
        ("flow_projection_fwd (no fill: the reference CPU code has none)", lambda: O.flow_projection_forward(f, 0)),
        ("flow_projection_fwd + restated fill-hole", lambda: O.flow_projection_forward(f, 1)),
        ("depth_flow_projection_fwd", lambda: O.depth_flow_projection_forward(f, d, 0)),
        ("flow_projection_bwd", lambda: O.flow_projection_backward(f, cnt, gf)),
        ("depth_flow_projection_bwd", lambda: O.depth_flow_projection_backward(f, d, dc, do, gf)),
        ("interpolation_fwd", lambda: O.interpolation_ch_forward(x, f)),
        ("interpolation_bwd", lambda: O.interpolation_ch_backward(x, f, g)),
    ]
    for name, fn in cases:
        fn()
        reps, spent = 0, 0.0
        while spent < 2.0 and reps < 50:
            t0 = time.perf_counter(); fn(); spent += time.perf_counter() - t0; reps += 1
        mp = B * H * W * reps / spent / 1e6
        rows.append({"op": "cpu_baseline " + name, "mpix_s": round(mp, 2), "threads": O.num_threads(),
                     "sample": "%dx%dx%dx%d, %d reps" % (B, C, H, W, reps)})
        print("%-72s %10.2f Mpix/s  (oracle, %d threads)" % ("cpu_baseline " + name, mp, O.num_threads()), flush=True)




def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--json", default=os.path.join(ROOT, "gpurun_out", "baselines.json"))
    ap.add_argument("--no-cpu", action="store_true")
    a = ap.parse_args()
    rows = []
    reference_gpu_rows(rows)
    if not a.no_cpu:
        cpu_rows(rows)
    os.makedirs(os.path.dirname(a.json), exist_ok=True)
    json.dump({"device": torch.cuda.get_device_name(0), "rows": rows}, open(a.json, "w"), indent=1)
    print("wrote", a.json)


if __name__ == "__main__":
    main()
