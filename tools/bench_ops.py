#!/usr/bin/env python
"""tools/bench_ops.py -- per-operator / per-variant measurement on one MI355X (HIP events around each launch).

Not the headline bench (that is bench.py); this is the A/B harness used while tuning and the source of the
secondary rows in DESIGN.md: every operator of the hot path at the BASELINE.json config sizes, with the
achieved ALGORITHMIC bandwidth (bytes each tensor touched once / median launch time) against the 8 TB/s HBM
peak.  Variants of one kernel are interleaved in one process (round-robin rounds) so their ratio is reliable.

    python tools/bench_ops.py [--quick] [--only fi_fwd,proj] [--json gpurun_out/bench_ops.json]
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

import my_package._ext.my_lib as L  # noqa: E402
from tools import measure  # noqa: E402


class _Knobs(object):
    """Variant selection.  The sweep runs on the PRODUCT library unless A/B or ablation variants are asked for
    (--variants / --proj-variants / --bwd-variants): those exist in the measurement build only (tools/measure.py),
    which then serves every row of the run."""
    active = False

    def enable(self):
        if not self.active:
            measure.use()
            self.active = True

    def set_variant(self, op, v):
        if self.active:
            measure.set_variant(op, v)
        elif int(v) != -1:
            raise RuntimeError("variant %s=%d needs the measurement build (call M.enable() first)" % (op, v))


M = _Knobs()
from tools import synth  # noqa: E402

PEAK = 8.0e12


def time_launches(fn, pre=None, warmup=12, iters=15, burst=1):
    """median / min launch time in seconds; `pre` (e.g. zero-filling outputs) runs outside the timed span.
    burst > 1: that many launches back to back between the two events, time divided by it -- for launches of a
    few tens of microseconds, where one launch between two events mostly measures the host's enqueue latency
    (the accumulating passes then add into non-zero buffers, which costs the same)."""
    for _ in range(warmup):
        if pre:
            pre()
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        if pre:
            pre()
        a = torch.cuda.Event(enable_timing=True)
        b = torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(burst):
            fn()
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b) * 1e-3 / burst)
    return statistics.median(ts), min(ts)


def report(rows, name, sites, bytes_per_site, med, mn, extra=None):
    alg = sites * bytes_per_site
    row = {"op": name, "sites": sites, "bytes_per_site": bytes_per_site, "median_us": round(med * 1e6, 1),
           "min_us": round(mn * 1e6, 1), "mpix_s": round(sites / med / 1e6, 1),
           "alg_GBps": round(alg / med / 1e9, 1), "hbm_frac": round(alg / med / PEAK, 4)}
    if extra:
        row.update(extra)
    rows.append(row)
    print("%-58s %9.1f us  %10.1f Mpix/s  %8.1f GB/s  %5.1f%% of 8 TB/s" %
          (name, row["median_us"], row["mpix_s"], row["alg_GBps"], 100 * row["hbm_frac"]), flush=True)


def bench_fi_fwd(rows, dev, B, C, H, W, flow_kind, variants, tag, rounds=4):
    """variants are timed in interleaved rounds in one process (the device needs ~100 launches to reach steady
    clocks, so whichever variant runs first would otherwise look ~5 % slower); median over all rounds."""
    # a launch smaller than the 256 MiB Infinity Cache rotates over input sets (> 1 GB cycled): the row is an HBM number
    nbytes = B * H * W * 4 * (2 * C + 2 + 16)
    n_sets = 1 if nbytes > 3e8 else int(1.0e9 // nbytes) + 1
    sets = [synth.torch_inputs(dev, B, C, H, W, flow_kind=flow_kind, seed=1234 + 97 * i) for i in range(n_sets)]
    outs = [torch.zeros_like(t["x"]) for t in sets]
    out = outs[0]
    turn = [0]

    def fn():
        i = turn[0] % n_sets
        turn[0] += 1
        return L.FilterInterpolationLayer_gpu_forward(sets[i]["x"], sets[i]["flow"], sets[i]["filt"], outs[i])
    M.set_variant("fi_fwd", variants[0])
    for _ in range(max(20, int(0.08 / max(1e-5, time_launches(fn, warmup=1, iters=3)[0])))):   # ~80 ms pre-warm
        fn()
    samples = {v: [] for v in variants}
    same = {}
    ref = None
    for _ in range(rounds if len(variants) > 1 else 1):
        for v in variants:
            M.set_variant("fi_fwd", v)
            for _ in range(2):
                fn()
            torch.cuda.synchronize()
            burst = 20 if B * H * W < 4e6 else 1           # tiny launches: see time_launches
            for _ in range(8):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(burst):
                    fn()
                b.record(); b.synchronize()
                samples[v].append(a.elapsed_time(b) * 1e-3 / burst)
            turn[0] = 0
            fn()                                            # (set 0 again: the comparison is on its output)
            if ref is None:
                ref = out.clone()
            same[v] = bool((out - ref).abs().max().item() <= 1e-5)
    for v in variants:
        report(rows, "fi_fwd %s C=%d %dx%dx%d flow=%s variant=%d%s" % (tag, C, B, H, W, flow_kind, v,
                                                                       " (bursts of 20, %d input sets)" % n_sets if B * H * W < 4e6 else ""),
               B * H * W, 4 * (2 * C + 2 + 16), statistics.median(samples[v]), min(samples[v]),
               {"variant": v, "matches_first_variant": same[v]})
    M.set_variant("fi_fwd", -1)


def bench_fi_blend(rows, dev, B, H, W, flow_kind):
    """fused dual warp + blend (188 B/site) against its composition: two forwards + the torch blend (236 B/site)"""
    a = synth.torch_inputs(dev, B, 3, H, W, flow_kind=flow_kind, seed=11)
    b = synth.torch_inputs(dev, B, 3, H, W, flow_kind=flow_kind, seed=12)
    o0, o1 = torch.rand(B, 1, H, W, device=dev), torch.rand(B, 1, H, W, device=dev)
    out = torch.empty_like(a["x"])
    w0, w2 = torch.zeros_like(a["x"]), torch.zeros_like(a["x"])
    med, mn = time_launches(lambda: L.FilterInterpolationBlendLayer_gpu_forward(
        a["x"], b["x"], a["flow"], b["flow"], a["filt"], b["filt"], o0, o1, out))
    report(rows, "fi_blend fused  C=3 %dx%dx%d flow=%s" % (B, H, W, flow_kind), B * H * W, 188, med, mn)

    def composed():
        L.FilterInterpolationLayer_gpu_forward(a["x"], a["flow"], a["filt"], w0)
        L.FilterInterpolationLayer_gpu_forward(b["x"], b["flow"], b["filt"], w2)
        torch.add(o0 * w0, o1 * w2, out=out)
    med, mn = time_launches(composed)
    report(rows, "fi_blend composed (2 fwd + torch blend) %dx%dx%d" % (B, H, W), B * H * W, 188, med, mn)


def bench_fi_ctx(rows, dev, B, C, H, W, flow_kind):
    """the warp stage of MEMC_Net_star for one frame pair (blended frame + both warped context tensors): one launch
    per direction with shared flow / tap reads (section 8f-3) against the fused blend + two context warps"""
    a = synth.torch_inputs(dev, B, 3, H, W, flow_kind=flow_kind, seed=11)
    b = synth.torch_inputs(dev, B, 3, H, W, flow_kind=flow_kind, seed=12)
    c0, c2 = torch.rand(B, C, H, W, device=dev), torch.rand(B, C, H, W, device=dev)
    o0, o1 = torch.rand(B, 1, H, W, device=dev), torch.rand(B, 1, H, W, device=dev)
    w0, out = torch.empty_like(a["x"]), torch.empty_like(a["x"])
    c0w, c2w = torch.empty_like(c0), torch.empty_like(c2)
    sites = B * H * W

    def fused():
        L.FilterInterpolationCtxLayer_gpu_forward(a["x"], c0, a["flow"], a["filt"], None, None, None, w0, c0w)
        L.FilterInterpolationCtxLayer_gpu_forward(b["x"], c2, b["flow"], b["filt"], w0, o0, o1, out, c2w)
    for _ in range(40):
        fused()
    med, mn = time_launches(fused)
    alg = 188 + 2 * 8 * C                   # compulsory bytes per site of the whole stage (each tensor once)
    report(rows, "warp stage C=3+%d fused per direction (2 launches) %dx%dx%d flow=%s" % (C, B, H, W, flow_kind),
           sites, alg, med, mn)

    med, mn = time_launches(lambda: L.FilterInterpolationCtxLayer_gpu_forward(
        a["x"], c0, a["flow"], a["filt"], None, None, None, w0, c0w))
    report(rows, "  image + context warp, one direction, no blend %dx%dx%d" % (B, H, W), sites, 4 * (2 * C + 6 + 18), med, mn)
    med, mn = time_launches(lambda: L.FilterInterpolationCtxLayer_gpu_forward(
        b["x"], c2, b["flow"], b["filt"], w0, o0, o1, out, c2w))
    report(rows, "  image + context warp, one direction, blend epilogue %dx%dx%d" % (B, H, W), sites,
           4 * (2 * C + 6 + 18 + 5), med, mn)
    med, mn = time_launches(lambda: L.FilterInterpolationLayer_gpu_forward(c0, a["flow"], a["filt"], c0w))
    report(rows, "  context warp alone (fi_fwd_tiled_c4n) %dx%dx%d" % (B, H, W), sites, 4 * (2 * C + 18), med, mn)

    def separate():
        L.FilterInterpolationBlendLayer_gpu_forward(a["x"], b["x"], a["flow"], b["flow"], a["filt"], b["filt"], o0, o1, out)
        L.FilterInterpolationLayer_gpu_forward(c0, a["flow"], a["filt"], c0w)
        L.FilterInterpolationLayer_gpu_forward(c2, b["flow"], b["filt"], c2w)
    med, mn = time_launches(separate)
    report(rows, "warp stage C=3+%d blend + 2 context warps (3 launches) %dx%dx%d flow=%s" % (C, B, H, W, flow_kind),
           sites, alg, med, mn)


def bench_fi_bwd(rows, dev, B, C, H, W, flow_kind, tag, variants=()):
    nbytes = B * H * W * 4 * (3 * C + 2 * (2 + 16))
    n_sets = 1 if nbytes > 3e8 else int(1.0e9 // nbytes) + 1       # (see bench_fi_fwd: cold launches for small shapes)
    sets = [synth.torch_inputs(dev, B, C, H, W, flow_kind=flow_kind, seed=1234 + 97 * i, with_grad=True) for i in range(n_sets)]
    grads = [(torch.zeros_like(t["x"]), torch.zeros_like(t["flow"]), torch.zeros_like(t["filt"])) for t in sets]
    t = sets[0]
    x, f, k, g = t["x"], t["flow"], t["filt"], t["gout"]
    g1, g2, g3 = grads[0]
    turn = [0]

    def rot():
        i = turn[0] % n_sets
        turn[0] += 1
        s_, (a1, a2, a3) = sets[i], grads[i]
        return L.FilterInterpolationLayer_gpu_backward(s_["x"], s_["flow"], s_["filt"], s_["gout"], a1, a2, a3)

    def pre():
        if n_sets == 1:
            g1.zero_(); g2.zero_(); g3.zero_()
    burst = 20 if B * H * W < 4e6 else 1
    for v in variants:
        M.set_variant("fi_bwd", v)
        med, mn = time_launches(rot, pre, burst=burst)
        report(rows, "fi_bwd %s C=%d %dx%dx%d flow=%s ABLATION variant=%d" % (tag, C, B, H, W, flow_kind, v),
               B * H * W, 4 * (3 * C + 2 * (2 + 16)), med, mn)
    M.set_variant("fi_bwd", -1)
    med, mn = time_launches(rot, pre, burst=burst)
    report(rows, "fi_bwd %s C=%d %dx%dx%d flow=%s%s" % (tag, C, B, H, W, flow_kind,
                                                       " (bursts of 20, %d input sets)" % n_sets if burst > 1 else ""),
           B * H * W, 4 * (3 * C + 2 * (2 + 16)), med, mn)
    if n_sets > 1:                                          # what rounds 1-4 reported: one set, out of the Infinity Cache
        med, mn = time_launches(lambda: L.FilterInterpolationLayer_gpu_backward(x, f, k, g, g1, g2, g3), None, burst=burst)
        report(rows, "fi_bwd %s C=%d %dx%dx%d flow=%s (bursts of 20, ONE input set: cache-warm)" % (tag, C, B, H, W, flow_kind),
               B * H * W, 4 * (3 * C + 2 * (2 + 16)), med, mn)


def bench_projection(rows, dev, B, H, W, flow_kind, tag, proj_variants=()):
    t = synth.torch_inputs(dev, B, 3, H, W, flow_kind=flow_kind, with_depth=True)
    f, d = t["flow"], t["depth"]
    cnt = f.new_zeros((B, 1, H, W))
    out = torch.zeros_like(f)
    for _ in range(150):                        # the device needs ~100 launches (60 ms) to reach its steady clocks
        L.FlowProjectionLayer_gpu_forward(f, cnt, out, 0)
    gout = torch.rand_like(f)
    gin = torch.zeros_like(f)
    gd = torch.zeros_like(d)

    def pre():
        cnt.zero_(); out.zero_()
    for pv in proj_variants:
        M.set_variant("projection", pv)
        for fh in (0, 1):
            med, mn = time_launches(lambda: L.FlowProjectionLayer_gpu_forward(f, cnt, out, fh), pre)
            report(rows, "flow_projection_fwd %s %dx%dx%d flow=%s fillhole=%d A/B variant=%d" % (
                tag, B, H, W, flow_kind, fh, pv), B * H * W, 20, med, mn)
        med, mn = time_launches(lambda: L.DepthFlowProjectionLayer_gpu_forward(f, d, cnt, out, 1), pre)
        report(rows, "depth_flow_projection_fwd %s %dx%dx%d flow=%s fillhole=1 A/B variant=%d" % (
            tag, B, H, W, flow_kind, pv), B * H * W, 24, med, mn)
    if proj_variants:
        M.set_variant("projection", -8)
        med, mn = time_launches(lambda: L.FlowProjectionLayer_gpu_forward(f, cnt, out, 1), pre)
        report(rows, "flow_projection_fwd %s %dx%dx%d flow=%s fillhole=1 ABLATION holes detected, none filled" % (
            tag, B, H, W, flow_kind), B * H * W, 20, med, mn)
    M.set_variant("projection", -1)
    for fh in (0, 1):
        med, mn = time_launches(lambda: L.FlowProjectionLayer_gpu_forward(f, cnt, out, fh), pre)
        report(rows, "flow_projection_fwd %s %dx%dx%d flow=%s fillhole=%d" % (tag, B, H, W, flow_kind, fh),
               B * H * W, 20, med, mn)
    med, mn = time_launches(lambda: L.DepthFlowProjectionLayer_gpu_forward(f, d, cnt, out, 1), pre)
    report(rows, "depth_flow_projection_fwd %s %dx%dx%d flow=%s fillhole=1" % (tag, B, H, W, flow_kind),
           B * H * W, 24, med, mn)
    pre(); L.FlowProjectionLayer_gpu_forward(f, cnt, out, 0)
    med, mn = time_launches(lambda: L.FlowProjectionLayer_gpu_backward(f, cnt, gout, gin), lambda: gin.zero_())
    report(rows, "flow_projection_bwd %s %dx%dx%d flow=%s" % (tag, B, H, W, flow_kind), B * H * W, 28, med, mn)
    pre(); L.DepthFlowProjectionLayer_gpu_forward(f, d, cnt, out, 0)

    def pre2():
        gin.zero_(); gd.zero_()
    med, mn = time_launches(lambda: L.DepthFlowProjectionLayer_gpu_backward(f, d, cnt, out, gout, gin, gd), pre2)
    # flow 8 + depth 4 + count 4 + forward output 8 + gradoutput 8 | gradinput1 8 + gradinput2 4 (rounds 1-3 booked 48)
    report(rows, "depth_flow_projection_bwd %s %dx%dx%d flow=%s" % (tag, B, H, W, flow_kind), B * H * W, 44, med, mn)


def bench_interp(rows, dev, B, C, H, W, flow_kind, tag):
    t = synth.torch_inputs(dev, B, C, H, W, flow_kind=flow_kind, with_grad=True)
    x, f, g = t["x"], t["flow"], t["gout"]
    out = torch.zeros_like(x)
    med, mn = time_launches(lambda: L.InterpolationChLayer_gpu_forward(x, f, out))
    report(rows, "interpolation_fwd %s C=%d %dx%dx%d flow=%s" % (tag, C, B, H, W, flow_kind), B * H * W,
           4 * (2 * C + 2), med, mn)
    g1, g2 = torch.zeros_like(x), torch.zeros_like(f)

    def pre():
        g1.zero_(); g2.zero_()
    med, mn = time_launches(lambda: L.InterpolationChLayer_gpu_backward(x, f, g, g1, g2), pre)
    report(rows, "interpolation_bwd %s C=%d %dx%dx%d flow=%s" % (tag, C, B, H, W, flow_kind), B * H * W,
           4 * (3 * C + 4), med, mn)


def bench_prologue(rows, dev, B, h, w):
    """FlowProjection's prologue in the networks: (20 * flow) / 2, x4 bilinear upsampling (MEMC_Net_star.py:172-176)"""
    import torch.nn.functional as F
    flow = torch.randn(B, 2, h, w, device=dev)
    out = torch.empty(B, 2, 4 * h, 4 * w, device=dev)
    sites = B * 16 * h * w
    med, mn = time_launches(lambda: L.FlowUpsample4Layer_gpu_forward(flow, out, 20.0, 2.0, False), warmup=40)
    report(rows, "flow prologue fused (x20 / 2, x4 bilinear) -> %dx2x%dx%d" % (B, 4 * h, 4 * w), sites, 8.5, med, mn)
    med, mn = time_launches(lambda: F.interpolate(20 * flow / 2.0, scale_factor=4, mode="bilinear", align_corners=False),
                            warmup=40)
    report(rows, "flow prologue torch (mul, div, interpolate) -> %dx2x%dx%d" % (B, 4 * h, 4 * w), sites, 8.5, med, mn)


def bench_copy(rows, dev):
    """calibration: a plain device copy of the same byte volume as the headline launch"""
    n = 2831155200 // 8
    a = torch.empty(n, dtype=torch.float32, device=dev).normal_()
    b = torch.empty_like(a)
    med, mn = time_launches(lambda: b.copy_(a))
    report(rows, "calibration: torch copy_ 1.42 GB read + 1.42 GB write", n, 8, med, mn)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--headline-only", action="store_true", help="fi_fwd: only the 720p batch-32 smooth-flow row")
    ap.add_argument("--ctx-only", action="store_true", help="fi_fwd: only the C=64 context-warp row")
    ap.add_argument("--ctx-flows", action="store_true", help="with --ctx-only: also the 4x smoother ('video') and the i.i.d. flow")
    ap.add_argument("--only", default="")
    ap.add_argument("--json", default=os.path.join(ROOT, "gpurun_out", "bench_ops.json"))
    ap.add_argument("--variants", default="-1", help="fi_fwd A/B arms, e.g. -1,1,0 (measurement build)")
    ap.add_argument("--proj-variants", default="")
    ap.add_argument("--bwd-variants", default="", help="fi_bwd ablation arms (timing only, wrong results)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    only = set(filter(None, args.only.split(",")))
    variants = [int(v) for v in args.variants.split(",")]
    rows = []
    if args.variants != "-1" or args.proj_variants or args.bwd_variants:
        M.enable()

    def want(k):
        return not only or k in only
    print(measure.version() if M.active else L.version(), torch.cuda.get_device_name(0), flush=True)
    if want("copy"):
        bench_copy(rows, dev)
    if want("fi_fwd") and args.ctx_only:
        bench_fi_fwd(rows, dev, 8, 64, 720, 1280, "smooth", variants, "ctx64")
        if args.ctx_flows:                         # the same kernel on the two other flow statistics
            bench_fi_fwd(rows, dev, 8, 64, 720, 1280, "video", variants[:1], "ctx64")
            bench_fi_fwd(rows, dev, 8, 64, 720, 1280, "iid", variants[:1], "ctx64")
    elif want("fi_fwd"):
        bench_fi_fwd(rows, dev, 32, 3, 720, 1280, "smooth", variants, "c_headline")
        if not args.headline_only:
            bench_fi_fwd(rows, dev, 32, 3, 720, 1280, "iid", variants[:1], "c_headline")
            bench_fi_fwd(rows, dev, 32, 3, 720, 1280, "video", variants[:1], "c_headline")
            bench_fi_fwd(rows, dev, 8, 3, 256, 448, "smooth", variants[:1], "c2")
        if not args.quick and not args.headline_only:
            bench_fi_fwd(rows, dev, 8, 64, 720, 1280, "smooth", variants[:1], "ctx64")
            bench_fi_fwd(rows, dev, 8, 3, 2160, 3840, "smooth", variants[:1], "c5_4k")
    if want("fi_blend"):
        bench_fi_blend(rows, dev, 32, 720, 1280, "smooth")
    if want("prologue"):
        bench_prologue(rows, dev, 32, 180, 320)
    if want("fi_ctx"):
        bench_fi_ctx(rows, dev, 8, 64, 720, 1280, "smooth")
    if want("fi_bwd"):
        bench_fi_bwd(rows, dev, 8, 3, 256, 448, "smooth", "c2", [int(v) for v in args.bwd_variants.split(",") if v])
        if not args.quick:
            bench_fi_bwd(rows, dev, 32, 3, 720, 1280, "smooth", "720p",
                         [int(v) for v in args.bwd_variants.split(",") if v])
            bench_fi_bwd(rows, dev, 32, 3, 720, 1280, "iid", "720p", [int(v) for v in args.bwd_variants.split(",") if v])
            bench_fi_bwd(rows, dev, 32, 3, 720, 1280, "video", "720p", [int(v) for v in args.bwd_variants.split(",") if v])
    if want("fi_bwd_ctx"):
        # (measurement build: + variant 40, the direct global-atomics kernel this path replaced -- 163 ms)
        bench_fi_bwd(rows, dev, 8, 64, 720, 1280, "smooth", "ctx64", [40] if M.active else [])
        bench_fi_bwd(rows, dev, 8, 64, 720, 1280, "iid", "ctx64")
        bench_fi_bwd(rows, dev, 4, 64, 256, 448, "smooth", "ctx64 crop")
    if want("proj"):
        bench_projection(rows, dev, 32, 720, 1280, "smooth", "c3",
                         [int(v) for v in args.proj_variants.split(",") if v])
        if not args.quick:
            pv2 = [int(v) for v in args.proj_variants.split(",") if v]
            bench_projection(rows, dev, 32, 720, 1280, "iid", "c3", pv2)
            bench_projection(rows, dev, 32, 720, 1280, "video", "c3")
    if want("interp"):
        bench_interp(rows, dev, 32, 3, 720, 1280, "smooth", "720p")
    if want("interp_ctx"):
        bench_interp(rows, dev, 8, 64, 720, 1280, "smooth", "ctx64")
        if M.active:                               # A/B: the direct global-atomics backward it replaced
            M.set_variant("bl_bwd_direct", 1)
            bench_interp(rows, dev, 8, 64, 720, 1280, "smooth", "ctx64 DIRECT backward")
            M.set_variant("bl_bwd_direct", 0)
    # baselines (the CPU oracle, the reference's own kernels on this GPU) live under tests/: tests/bench_baselines.py
    os.makedirs(os.path.dirname(args.json), exist_ok=True)
    json.dump({"device": torch.cuda.get_device_name(0), "lib": measure.version() if M.active else L.version(), "rows": rows}, open(args.json, "w"),
              indent=1)
    print("wrote", args.json)


if __name__ == "__main__":
    main()
