#!/usr/bin/env python
"""bench.py -- headline benchmark of the adaptive-warp hot path on MI355X.

Metric (BASELINE.json): Mpixels/s of the fused adaptive-warp (FilterInterpolation) forward, 4x4 filter, C=3,
fp32, batch 32 of 1280x720 frames per GPU, synthetic inputs already resident in HBM.  One "step" = one pass of
the operator over one batch (one kernel launch through the C ABI, on torch's current stream).

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Multi-GPU: frame pairs are independent, so the batch is sharded over the ranks and there is no collective on the
data path; RCCL is used only to broadcast the run configuration from rank 0 before the timed region and to take the
max of the per-rank times after it.  Default = STRONG scaling, as BASELINE.json / SURVEY.md section 8(e) state it:
the global batch stays 32 frame pairs and rank r owns the contiguous shard of 32 / N of them (32 / 16 / 8 / 4 per GPU
at N = 1 / 2 / 4 / 8); `--scaling weak` gives every rank its own batch of 32 instead (global batch 32 N).

Timing: W warm-up steps, then R (`--windows`, 7) windows of exactly K steps.  A window is  barrier + synchronize | t0 | K
launches | the rank's own synchronize | t1  -- no collective and no event inside the clock; `value` comes from the MEDIAN
window, each window's time being the max over the ranks (one all-reduce after the last window).  A shard of 4 frames is a
66 us launch and the host enqueues one in ~10 us: eager launches keep the queue full (`--launch graph` replays the K steps
from one captured HIP graph instead; it starts a 20-step window ~35 us later and is no faster at 300 steps).

Prints ONE JSON line on rank 0 (see README / DESIGN.md "Measurement" for the field definitions).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "memc-net_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

HBM_PEAK_BPS = 8.0e12           # MI355X HBM3E spec peak (guides/MI355X_MICROARCH.md: 8 TB/s; ~6.3 achievable)
BYTES_PER_SITE = {               # ALGORITHMIC bytes per output site, every tensor touched once
    "fi_fwd": lambda C, fs: 4 * (2 * C + 2 + fs * fs),
    "fi_bwd": lambda C, fs: 4 * (3 * C + 2 * (2 + fs * fs)),
    "fi_bwd_nog1": lambda C, fs: 4 * (2 * C + 2 * (2 + fs * fs)),     # gradinput1 not wanted (NULL): no image gradient written
    "proj_fwd": lambda C, fs: 20,
    "depth_proj_fwd": lambda C, fs: 24,
    "proj_bwd": lambda C, fs: 28,
    "depth_proj_bwd": lambda C, fs: 44,        # flow 8 + depth 4 + count 4 + forward output 8 + gradoutput 8 + gradinput1 8 + gradinput2 4
    "interp_fwd": lambda C, fs: 4 * (2 * C + 2),
    "interp_bwd": lambda C, fs: 4 * (3 * C + 4),
}
CHECK_TOLERANCE = 1e-4           # BASELINE.json: "outputs within 1e-4 of reference"


# ----------------------------------------------------------------------------------------------------------
# rank / sharding / timing plumbing (backend-agnostic so that it is testable with gloo on CPU)
# ----------------------------------------------------------------------------------------------------------
def dist_env():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def shard_plan(rank, world, batch, base_seed, scaling="weak"):
    """Independent frame pairs, contiguous shards, no halo, no exchange.
    weak:   every rank owns `batch` items of a global batch of world * batch (item g lives on rank g // batch).
    strong: the global batch IS `batch`; rank r owns items [first, first + items) with the remainder spread over
            the first ranks (32 over 8 ranks: 4 each; 32 over 3 ranks: 11, 11, 10)."""
    if scaling == "weak":
        first, items, total = rank * batch, batch, world * batch
    elif scaling == "strong":
        q, r = divmod(batch, world)
        items = q + (1 if rank < r else 0)
        first = rank * q + min(rank, r)
        total = batch
    else:
        raise ValueError(scaling)
    return {"rank": rank, "world": world, "first_item": first, "items": items, "global_batch": total,
            "seed": base_seed + rank, "scaling": scaling}


# `--dist-single` (a one-GPU box): the WHOLE collective layer runs on a real RCCL communicator of one rank -- communicator
# creation with device_id, the broadcast, the barriers, the float64 MAX all-reduce, the all-gather -- instead of being skipped
# for world == 1.  It is the closest a one-GPU lease gets to the N > 1 path over RCCL; the line's `dist` record says so.
_FORCE_DIST = [False]


def _dist_on(world):
    return world > 1 or _FORCE_DIST[0]


def broadcast_config(cfg, world, device):
    """Rank 0's run configuration wins (RCCL broadcast over xGMI on GPUs, gloo on CPU).  Ints only."""
    if not _dist_on(world):
        return cfg
    import torch
    import torch.distributed as dist
    keys = sorted(cfg)
    t = torch.tensor([int(cfg[k]) for k in keys], dtype=torch.int64, device=device)
    dist.broadcast(t, src=0)
    return {k: int(v) for k, v in zip(keys, t.tolist())}


def barrier_sync(world, device):
    import torch
    if _dist_on(world):
        import torch.distributed as dist
        dist.barrier()
    if device.type == "cuda":
        torch.cuda.synchronize(device)


def _drain(device, stream=None):
    """The closing synchronize of a window.  On a GPU the host first spins on the stream the steps were enqueued on (a
    hipStreamQuery loop sees completion within a microsecond or two; hipDeviceSynchronize alone may yield the core first),
    then calls torch.cuda.synchronize -- which is what the contract asks for and returns at once by then."""
    import torch
    if device.type == "cuda":
        if stream is not None:
            while not stream.query():
                pass
        torch.cuda.synchronize(device)


def timed_windows(window_fn, windows, world, device):
    """`windows` repetitions of ONE timed window.  A window is: barrier + synchronize (every rank starts together, outside
    the clock), t0, `window_fn(r)` (enqueues exactly K steps and returns the stream they went to, or None), this
    rank's own synchronize, t1.  Nothing collective sits inside [t0, t1]: the ranks never talk on the data path, so a
    rank's window is over when ITS device is idle -- a closing dist.barrier() inside the clock would add the
    collective's latency (and, at the driver's 20 steps of a 66 us shard, 10-20 % of the window) to every rank's time.
    The max over ranks is taken AFTER the last window, in one all_reduce(MAX) over the vector of local window times.
    Returns (per-window max-over-ranks seconds, this rank's per-window seconds)."""
    import torch
    local = []
    enqueue = []                                          # host time to enqueue the window's work (evidence: GPU-bound or not)
    for r in range(windows):
        barrier_sync(world, device)
        t0 = time.perf_counter()
        last = window_fn(r)
        t_enq = time.perf_counter()
        _drain(device, last)
        local.append(time.perf_counter() - t0)
        enqueue.append(t_enq - t0)
    timed_windows.last_enqueue_s = enqueue
    worst = list(local)
    if _dist_on(world):
        import torch.distributed as dist
        t = torch.tensor(local, dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)          # the closing collective: outside every window
        worst = [float(v) for v in t.tolist()]
    return worst, local


def median(values):
    v = sorted(values)
    n = len(v)
    return v[n // 2] if n % 2 else 0.5 * (v[n // 2 - 1] + v[n // 2])


def timed_steps(step_fn, steps, warmup, world, device, windows=1):
    """`warmup` untimed steps, then `windows` windows of exactly `steps` steps each (timed_windows); step_fn gets the step
    index inside a window, None while warming up.  Returns (median over windows of the max-over-ranks wall seconds of
    one window, this rank's median)."""
    for _ in range(warmup):
        step_fn(None)

    def window(_r):
        for i in range(steps):
            step_fn(i)
        return None
    worst, local = timed_windows(window, windows, world, device)
    return median(worst), median(local)


def barrier_cost_us(world, device, reps=9):
    """Evidence only: what one empty dist.barrier() costs on this job (median of `reps`), i.e. what every window would
    carry if the barrier were inside the clock."""
    if not _dist_on(world):
        return None
    import torch.distributed as dist
    barrier_sync(world, device)
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        dist.barrier()
        ts.append(time.perf_counter() - t0)
    return round(median(ts) * 1e6, 1)


# ----------------------------------------------------------------------------------------------------------
# CPU baseline (rank 0, N = 1 only): the oracle -- a port of the reference's CPU code -- on a bounded sample
# ----------------------------------------------------------------------------------------------------------
def cpu_baseline(x, flow, filt, target_seconds=12.0):
    """Times oracle.filter_interpolation_forward on the first few frames of the SAME tensors the GPU ran on.
    Returns (the cpu_baseline object, the oracle's output for those frames) -- the second is what the GPU result of
    the timed launches is checked against."""
    import numpy as np
    from oracle import memc_oracle as O          # checker / baseline only -- never on the product path
    O.build()
    cores = O.num_threads()
    nb = min(x.shape[0], 4)
    xs, fs_, ks = (t[:nb].cpu().numpy() for t in (x, flow, filt))
    sites = nb * xs.shape[2] * xs.shape[3]
    O.filter_interpolation_forward(xs[:1], fs_[:1], ks[:1])     # page in
    reps, spent, want = 0, 0.0, None
    while spent < target_seconds and reps < 200:
        t0 = time.perf_counter()
        want = O.filter_interpolation_forward(xs, fs_, ks)
        spent += time.perf_counter() - t0
        reps += 1
    assert np.isfinite(spent)
    return {"value": round(sites * reps / spent / 1e6, 2), "unit": "Mpixels/s", "cores": cores, "kind": "port",
            "sample": "oracle FilterInterpolation fwd, first %d frames of the GPU batch (%dx%dx%dx%d), %d reps, "
                      "%.1f s, OpenMP over batch x rows" % (nb, nb, xs.shape[1], xs.shape[2], xs.shape[3], reps, spent)}, want


def check_against_oracle(out, want):
    """The output of the timed launches (first frames) against the oracle result the cpu_baseline leg computed."""
    import numpy as np
    nb = want.shape[0]
    got = out[:nb].cpu().numpy()
    err = float(np.abs(got.astype(np.float64) - want.astype(np.float64)).max())
    return {"max_abs_err": err, "frames": int(nb), "tolerance": CHECK_TOLERANCE, "ok": bool(err <= CHECK_TOLERANCE),
            "against": "oracle (CPU restatement of my_lib.c), same input tensors as the timed launches"}


def kernel_source_hash():
    """sha256 over the HIP sources of the dominant kernel: the PMC traffic record is only quoted for the very code
    it was measured on (tools/pmc_traffic.py stores the same hash)."""
    import hashlib
    h = hashlib.sha256()
    for name in ("filter_interpolation.hip", "memc_tile.hpp", "memc_common.hpp"):
        with open(os.path.join(ROOT, "memc-net_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def load_traffic(workload_key):
    """HBM bytes per launch from the committed PMC profile of this same command (profiles/traffic.json, written by
    tools/pmc_traffic.py in separate FETCH_SIZE / WRITE_SIZE passes); None when there is no record for this
    workload or the kernel sources changed since it was taken (stale counters are not quoted)."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        data = json.load(open(path))
        rec = data.get(workload_key)
        if rec is None or rec.get("kernel_source_hash") != kernel_source_hash():
            return None, None
        return rec.get("hbm_bytes_per_launch"), "profiles/traffic.json@%s" % rec.get("kernel_source_hash")
    except (OSError, ValueError):
        return None, None


def copy_calibration(my_lib, device, nbytes, iters=20):
    """What THIS box's HBM gives the access pattern of the tiled kernels (memc_calibration_stream, include/memc_warp.h:
    16 bytes per lane, non-temporal, the XCD walk), by HIP events, for the same byte volume as the headline launch:
    a copy (1 read : 1 write) and the adaptive warp's own mix (7 reads : 1 write).  Rounds 2-3 calibrated on torch's
    copy_, which the headline kernel outran; `roofline.achievable_peak` is now the better of these two rates, next
    to the 8 TB/s specification."""
    import torch
    rates = {}
    for name, reads in (("copy_1r1w", 1), ("mix_7r1w", 7)):
        n4 = nbytes // (16 * (reads + 1))                  # float4 per stream
        src = torch.empty(reads * n4 * 4, dtype=torch.float32, device=device).normal_()
        dst = torch.empty(n4 * 4, dtype=torch.float32, device=device)
        for _ in range(5):
            if my_lib.calibration_stream(src, dst, reads) != 0:
                raise RuntimeError("memc_calibration_stream failed")
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
        for s0, s1 in ev:
            s0.record(); my_lib.calibration_stream(src, dst, reads); s1.record()
        torch.cuda.synchronize(device)
        ts = sorted(s0.elapsed_time(s1) for s0, s1 in ev)
        rates[name] = 16.0 * n4 * (reads + 1) / (ts[len(ts) // 2] * 1e-3)
        del src, dst
    return rates


def _avg_launch_s(fn, torch, device, warm=15, iters=40, burst=1, warm_seconds=0.08):
    # warm-up by TIME as well as by count: after host-side work (allocations, random fills) the device needs ~60 ms of
    # launches to be back at its steady clocks (tools/timeline.py) -- 15 launches of a 120 us kernel are 2 ms, and the
    # first secondary row measured behind such a pause read 10 % slow (round 4, config 3 without hole filling)
    for _ in range(warm):
        fn()
    torch.cuda.synchronize(device)
    t_end = time.perf_counter() + warm_seconds
    while time.perf_counter() < t_end:
        for _ in range(8):
            fn()
        torch.cuda.synchronize(device)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a0, a1 in ev:
        a0.record()
        for _ in range(burst):
            fn()
        a1.record()
    torch.cuda.synchronize(device)
    return sum(a0.elapsed_time(a1) for a0, a1 in ev) / len(ev) / burst / 1e3


def secondary_rows(my_lib, synth, torch, device, seed):
    """The other BASELINE configs, OUTSIDE the timed region (rank 0, N = 1): per-launch HIP-event averages and the
    fraction of the 8 TB/s HBM peak their ALGORITHMIC bytes amount to.  Launches shorter than ~0.2 ms are timed in
    bursts between two events (one launch between two events mostly measures the host's enqueue cost): every row says how
    in its "timing" field ("single" | "burst N").  Launches that move less than the 256 MiB Infinity Cache (config 2) rotate
    over input sets -- avg_launch_us is the COLD figure, cache_warm_us the single-set one rounds 1-4 reported."""
    rows = {}

    def row(name, op, C, sites, seconds, note=None, timing="single"):
        nbytes = BYTES_PER_SITE[op](C, 4) * sites
        rows[name] = {"avg_launch_us": round(seconds * 1e6, 2), "mpixels_s": round(sites / seconds / 1e6, 1),
                      "algorithmic_bytes_per_launch": nbytes, "frac": round(nbytes / seconds / HBM_PEAK_BPS, 4),
                      "timing": timing}
        if note:
            rows[name]["note"] = note

    def rotating(calls):
        """One callable that takes the next of `calls` at every invocation: consecutive launches work on different input
        sets, so that a launch smaller than the 256 MiB Infinity Cache cannot be served from it."""
        state = [0]

        def fn():
            calls[state[0] % len(calls)]()
            state[0] += 1
        return fn

    def small_row(name, op, C, sites, make_call, n_sets, note=None, burst=20):
        """A launch of tens of microseconds on tens of megabytes: timed COLD (rotating over n_sets input sets, the figure
        the row reports: an HBM number) and cache-warm (one set relaunched: what rounds 1-4 reported), bursts of `burst`."""
        calls = [make_call(i) for i in range(n_sets)]
        cold = _avg_launch_s(rotating(calls), torch, device, burst=burst)
        warm = _avg_launch_s(calls[0], torch, device, burst=burst)
        row(name, op, C, sites, cold, note, timing="burst %d" % burst)
        rows[name]["input_sets"] = n_sets
        rows[name]["cache_warm_us"] = round(warm * 1e6, 2)
        rows[name]["cache_warm_frac"] = round(BYTES_PER_SITE[op](C, 4) * sites / warm / HBM_PEAK_BPS, 4)

    # config 2: fused adaptive warp fwd + bwd, 448 x 256 (Vimeo septuplet), batch 8.  One launch moves 88 MB (forward) /
    # 165 MB (backward): relaunched on one input set it runs out of the 256 MiB Infinity Cache (round-4 review).  The rows
    # rotate over enough input sets to cycle > 1 GB; the single-set figure stays beside it as cache_warm_us.
    sites = 8 * 256 * 448
    n_fwd, n_bwd = 12, 7
    sets = [synth.torch_inputs(device, 8, 3, 256, 448, flow_kind="smooth", seed=seed + 2 + 97 * i, with_grad=True)
            for i in range(n_fwd)]
    outs = [torch.zeros_like(t["x"]) for t in sets]
    grads = [(torch.zeros_like(t["x"]), torch.zeros_like(t["flow"]), torch.zeros_like(t["filt"])) for t in sets[:n_bwd]]

    def fwd_call(i):
        t, out = sets[i], outs[i]
        return lambda: my_lib.FilterInterpolationLayer_gpu_forward(t["x"], t["flow"], t["filt"], out)

    def bwd_call(i):
        t, (g1, g2, g3) = sets[i], grads[i]
        return lambda: my_lib.FilterInterpolationLayer_gpu_backward(t["x"], t["flow"], t["filt"], t["gout"], g1, g2, g3)

    def bwd_nog1_call(i):
        t, (g1, g2, g3) = sets[i], grads[i]
        return lambda: my_lib.FilterInterpolationLayer_gpu_backward(t["x"], t["flow"], t["filt"], t["gout"], None, g2, g3)

    small_row("config2_fi_fwd_8x3x256x448", "fi_fwd", 3, sites, fwd_call, n_fwd)
    small_row("config2_fi_bwd_8x3x256x448", "fi_bwd", 3, sites, bwd_call, n_bwd)
    # ... and as the reference's networks run it: the warped frames are data, autograd does not ask for gradinput1
    # (MEMC_Net_star.py:266-277) -- the extension of include/memc_warp.h (gradinput1 NULL)
    small_row("config2_fi_bwd_without_image_gradient_8x3x256x448", "fi_bwd_nog1", 3, sites, bwd_nog1_call, n_bwd,
              "gradinput1 = NULL (extension): flow and tap gradients only")
    del sets, outs, grads
    # the backward at the headline size (the only channel count the reference back-propagates through)
    t = synth.torch_inputs(device, 32, 3, 720, 1280, flow_kind="smooth", seed=seed + 6, with_grad=True)
    g1, g2, g3 = torch.zeros_like(t["x"]), torch.zeros_like(t["flow"]), torch.zeros_like(t["filt"])
    row("fi_bwd_32x3x720x1280", "fi_bwd", 3, 32 * 720 * 1280, _avg_launch_s(
        lambda: my_lib.FilterInterpolationLayer_gpu_backward(t["x"], t["flow"], t["filt"], t["gout"], g1, g2, g3),
        torch, device))

    def bwd_with_memset():                      # what a caller pays: the RGB backward ADDS to gradinput1 (12 B/site to clear)
        g1.zero_()
        my_lib.FilterInterpolationLayer_gpu_backward(t["x"], t["flow"], t["filt"], t["gout"], g1, g2, g3)
    rows["fi_bwd_32x3x720x1280"]["with_memset_us"] = round(_avg_launch_s(bwd_with_memset, torch, device) * 1e6, 2)
    rows["fi_bwd_32x3x720x1280"]["note"] = ("with_memset_us: the same call preceded by the zero fill of gradinput1 it relies on "
                                            "(C = 3 accumulates; memc_gradinput1_is_stored)")
    row("fi_bwd_without_image_gradient_32x3x720x1280", "fi_bwd_nog1", 3, 32 * 720 * 1280, _avg_launch_s(
        lambda: my_lib.FilterInterpolationLayer_gpu_backward(t["x"], t["flow"], t["filt"], t["gout"], None, g2, g3),
        torch, device), "gradinput1 = NULL (extension): what a training step of the reference's networks needs")
    del g1, g2, g3
    # config 3: FlowProjection / DepthFlowProjection scatter, 1280 x 720, batch 32 (same flow; + depth)
    from my_package.modules.FlowProjectionModule import FlowProjectionModule
    from my_package.modules.DepthFlowProjectionModule import DepthFlowProjectionModule
    f = t["flow"]
    del t
    dep = torch.rand((32, 1, 720, 1280), device=device) + 0.1
    cnt, po = torch.zeros((32, 1, 720, 1280), device=device), torch.zeros_like(f)
    sites = 32 * 720 * 1280
    # (637 / 755 MB per launch: beyond the Infinity Cache.  Back-to-back launches overlap one call's tail kernels with the next
    # call's start; single_call_us is one call between two events)
    for fill in (0, 1):
        name = "config3_flow_projection_fwd_fillhole%d_32x720x1280" % fill
        call = lambda: my_lib.FlowProjectionLayer_gpu_forward(f, cnt, po, fill)      # noqa: E731
        row(name, "proj_fwd", 0, sites, _avg_launch_s(call, torch, device, burst=4), timing="burst 4")
        rows[name]["single_call_us"] = round(_avg_launch_s(call, torch, device, warm=4, warm_seconds=0.0) * 1e6, 2)
        # ... and as the package runs it (my_package.modules.FlowProjectionModule: torch.empty outputs, a torch.empty
        # workspace, the `_ws` entry point, no block kept by the library), same bursts
        layer = FlowProjectionModule(requires_grad=(fill == 0))
        with torch.no_grad():
            rows[name]["layer_call_us"] = round(_avg_launch_s(lambda: layer(f), torch, device, burst=4) * 1e6, 2)
    name = "config3_depth_flow_projection_fwd_fillhole1_32x720x1280"
    call = lambda: my_lib.DepthFlowProjectionLayer_gpu_forward(f, dep, cnt, po, 1)   # noqa: E731
    row(name, "depth_proj_fwd", 0, sites, _avg_launch_s(call, torch, device, burst=4), timing="burst 4")
    rows[name]["single_call_us"] = round(_avg_launch_s(call, torch, device, warm=4, warm_seconds=0.0) * 1e6, 2)
    dlayer = DepthFlowProjectionModule(requires_grad=False)
    with torch.no_grad():
        rows[name]["layer_call_us"] = round(_avg_launch_s(lambda: dlayer(f, dep), torch, device, burst=4) * 1e6, 2)
    # ... and under LARGE motion (not a BASELINE config; sources that move 24 px or more take proj_owner_far):
    # the same flow twice as large (fast objects: a few per cent of the tiles are recomputed) and under a camera pan of
    # (40, -20) px (every source far, every tile recomputed, an uncovered band of holes along two edges)
    for tag, fl in (("motion_x2", f * 2.0), ("pan40", f + torch.tensor([40.0, -20.0], device=device).view(1, 2, 1, 1))):
        row("flow_projection_fwd_fillhole1_%s_32x720x1280" % tag, "proj_fwd", 0, sites, _avg_launch_s(
            lambda: my_lib.FlowProjectionLayer_gpu_forward(fl, cnt, po, 1), torch, device, burst=4), "large motion", timing="burst 4")
    del fl
    # ... and their backward passes (the count / output planes of a forward without hole filling, as in training)
    gout, gin, gd = torch.rand_like(f), torch.zeros_like(f), torch.zeros_like(dep)
    cnt.zero_(); po.zero_()
    my_lib.FlowProjectionLayer_gpu_forward(f, cnt, po, 0)
    row("config3_flow_projection_bwd_32x720x1280", "proj_bwd", 0, sites, _avg_launch_s(
        lambda: my_lib.FlowProjectionLayer_gpu_backward(f, cnt, gout, gin), torch, device, burst=4), timing="burst 4")
    cnt.zero_(); po.zero_()
    my_lib.DepthFlowProjectionLayer_gpu_forward(f, dep, cnt, po, 0)
    row("config3_depth_flow_projection_bwd_32x720x1280", "depth_proj_bwd", 0, sites, _avg_launch_s(
        lambda: my_lib.DepthFlowProjectionLayer_gpu_backward(f, dep, cnt, po, gout, gin, gd), torch, device, burst=4),
        timing="burst 4")
    del dep, cnt, po, gin, gd
    # the bilinear warp (Interpolation) at the headline size, forward and backward
    x = torch.rand((32, 3, 720, 1280), device=device)
    out, g1, g2 = torch.zeros_like(x), torch.zeros_like(x), torch.zeros_like(f)
    gx = torch.rand_like(x)
    row("interpolation_fwd_32x3x720x1280", "interp_fwd", 3, sites, _avg_launch_s(
        lambda: my_lib.InterpolationLayer_gpu_forward(x, f, out), torch, device, burst=4), timing="burst 4")
    row("interpolation_bwd_32x3x720x1280", "interp_bwd", 3, sites, _avg_launch_s(
        lambda: my_lib.InterpolationLayer_gpu_backward(x, f, gx, g1, g2), torch, device, burst=4), timing="burst 4")
    del f, gout, x, out, g1, g2, gx
    # config 5: 4K adaptive warp forward, batch 8
    t = synth.torch_inputs(device, 8, 3, 2160, 3840, flow_kind="smooth", seed=seed + 5)
    out = torch.zeros_like(t["x"])
    row("config5_fi_fwd_8x3x2160x3840", "fi_fwd", 3, 8 * 2160 * 3840, _avg_launch_s(
        lambda: my_lib.FilterInterpolationLayer_gpu_forward(t["x"], t["flow"], t["filt"], out), torch, device, iters=20))
    del t, out
    # the 64-channel context warp of config 4's network (MEMC_Net_star.py:281-285), batch 8
    t = synth.torch_inputs(device, 8, 64, 720, 1280, flow_kind="smooth", seed=seed + 4)
    out = torch.zeros_like(t["x"])
    row("context_warp_fi_fwd_8x64x720x1280", "fi_fwd", 64, 8 * 720 * 1280, _avg_launch_s(
        lambda: my_lib.FilterInterpolationLayer_gpu_forward(t["x"], t["flow"], t["filt"], out), torch, device, iters=20))
    del t, out
    torch.cuda.empty_cache()
    return rows


def config4_row(my_lib, torch, device, seed, pairs=4, height=720, width=1280, warm=2, steps=3):
    """BASELINE config 4 as one GPU of the 8-GPU job sees it: MEMC_Net_star inference (random weights, seeded) on its shard
    of 32 / 8 = 4 frame pairs of 1280 x 720, through networks/inference.py (pad to multiples of 128 the way
    demo_HD720p.py:88-118 does, network MEMC_Net_star.py:78-150, crop) on the HIP operators.  One untimed pass first (MIOpen
    picks its convolution solvers on first use), then `warm` warm-up and `steps` timed steps; hot_path_* are HIP-event spans
    around the operator entry points in one more pass (tools/bench_model.py).  The dense layers are stock torch.nn on
    MIOpen / rocBLAS: they are the reference's, not this repository's, and they are 99 % of the step."""
    import networks
    from tools.bench_model import instrumented_pass
    # MIOpen compiles (and times) its convolution kernels on first use -- 65 s for this network on a fresh box, whose image has no
    # kernel database.  memc-net_amd/networks/miopen_cache/ holds what such a first pass leaves behind (344 KB, see its README);
    # MIOpen gets a private copy, unless the caller already chose a cache directory.
    cache_src = os.path.join(ROOT, "memc-net_amd", "networks", "miopen_cache")
    cache = "caller's"
    if "MIOPEN_USER_DB_PATH" not in os.environ and "MIOPEN_CUSTOM_CACHE_DIR" not in os.environ:
        cache = "none"
        if os.path.isdir(cache_src):
            import shutil
            import tempfile
            tmp = tempfile.mkdtemp(prefix="memc_miopen_")
            for name in os.listdir(cache_src):
                if not name.endswith(".md"):
                    shutil.copy(os.path.join(cache_src, name), tmp)
            os.environ["MIOPEN_USER_DB_PATH"] = os.environ["MIOPEN_CUSTOM_CACHE_DIR"] = tmp
            cache = "in-tree copy"
    t_setup = time.perf_counter()
    torch.manual_seed(seed)
    with torch.device(device):
        net = networks.MEMC_Net_star(channel=3, filter_size=4, training=False)
    net = net.to(device).eval()
    g = torch.Generator(device=device).manual_seed(seed + 99)
    frames = torch.rand((2, pairs, 3, height, width), device=device, generator=g)

    def one_step():
        return networks.interpolate_pairs(net, frames[0], frames[1])
    with torch.no_grad():
        first = one_step()                               # MIOpen warm-up
        torch.cuda.synchronize(device)
        setup_s = time.perf_counter() - t_setup
        for _ in range(warm):
            one_step()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for a0, a1 in ev:
            a0.record()
            out = one_step()
            a1.record()
        torch.cuda.synchronize(device)
        wall = time.perf_counter() - t0
        per_op, n_calls, pass_ms = instrumented_pass(my_lib, torch, device, one_step)
    step_ms = sum(a0.elapsed_time(a1) for a0, a1 in ev) / steps
    hot_ms = sum(per_op.values())
    finite = bool(torch.isfinite(out).all()) and tuple(out.shape) == (pairs, 3, height, width)
    row = {"frames_per_s": round(pairs * steps / wall, 3), "ms_per_step": round(wall / steps * 1e3, 2),
           "gpu_ms_per_step": round(step_ms, 2), "pairs_per_step": pairs, "steps": steps, "warmup": warm,
           "hot_path_ms": round(hot_ms, 3), "hot_path_share": round(hot_ms / pass_ms, 4), "hot_path_calls": n_calls,
           "hot_path_ops_ms": {k: round(v, 3) for k, v in sorted(per_op.items())},
           "instrumented_pass_ms": round(pass_ms, 2), "setup_s": round(setup_s, 2), "miopen_cache": cache, "output_finite": finite,
           "repeatable": bool(torch.equal(first, out)),
           "note": "one GPU's shard (32 pairs / 8 GPUs) of config 4; random weights; dense layers = stock torch.nn (MIOpen), "
                   "hot path = this repository's HIP operators; frames_per_s counts interpolated frames"}
    del net, frames, out, first
    torch.cuda.empty_cache()
    return row


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--windows", type=int, default=7,
                    help="repetitions of the K-step timed window; `value` is the MEDIAN window (max over ranks per window)")
    ap.add_argument("--prewarm", type=int, default=200,
                    help="untimed launches before the warm-up steps: the device needs ~100 launches (60 ms) to "
                         "reach its steady clocks (tools/timeline.py); never part of the timed region")
    ap.add_argument("--batch", type=int, default=32, help="frame pairs: the GLOBAL batch (strong) / per GPU (weak)")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"])
    ap.add_argument("--launch", default="eager", choices=["auto", "eager", "graph"],
                    help="graph: the K steps of a window are one captured HIP graph replay (auto = eager: see main)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the rows outside the timed region (i.i.d. flow, the other BASELINE configs, copy calibration)")
    ap.add_argument("--input-sets", type=int, default=0,
                    help="input sets the launches rotate over (0 = automatic: enough to cycle >= 2.8 GB, what the full batch "
                         "moves per launch -- a smaller shard re-launched on the same tensors is partly served by the 256 MB "
                         "Infinity Cache, and its 'HBM fraction' would not be one)")
    ap.add_argument("--height", type=int, default=720)
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--channels", type=int, default=3)
    ap.add_argument("--flow", default="smooth", choices=["smooth", "iid"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU work the cpu_baseline leg is bounded to")
    # plumbing test of the N > 1 path on a box with fewer GPUs than ranks (ranks share devices, gloo instead of
    # RCCL, which refuses two ranks on one device): the line it prints is NOT a measurement
    ap.add_argument("--share-gpu", action="store_true", help=argparse.SUPPRESS)
    # one rank, but through torch.distributed on backend nccl (= RCCL): see _FORCE_DIST above
    ap.add_argument("--dist-single", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args(argv)

    import torch
    rank, local_rank, world = dist_env()
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs the torch.distributed.run launcher (WORLD_SIZE=%d)" % (args.gpus, world))
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    dev_index = local_rank % torch.cuda.device_count() if args.share_gpu else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.share_gpu:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=device)  # nccl == RCCL on ROCm
    elif args.dist_single:
        import socket
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        with socket.socket() as sock:           # a free port on the loopback for the one-rank rendezvous
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        dist.init_process_group(backend="nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=device)
        _FORCE_DIST[0] = True

    import my_package._ext.my_lib as my_lib                  # raises if libmemc_hip.so is missing
    from tools import synth

    cfg = broadcast_config({"batch": args.batch, "height": args.height, "width": args.width,
                            "channels": args.channels, "seed": 1234, "steps": args.steps,
                            "warmup": args.warmup, "windows": max(1, args.windows)}, world, device)
    C, H, W, fs = cfg["channels"], cfg["height"], cfg["width"], 4
    plan = shard_plan(rank, world, cfg["batch"], cfg["seed"], args.scaling)
    B = plan["items"]                           # this rank's shard
    if B == 0:
        raise SystemExit("rank %d has no frame pair (global batch %d over %d ranks)" % (rank, cfg["batch"], world))
    sites_per_launch = B * H * W
    alg_bytes = BYTES_PER_SITE["fi_fwd"](C, fs) * sites_per_launch
    # Input sets the launches rotate over: enough to cycle what the FULL batch moves in one launch (2.83 GB) -- one set at batch 32,
    # 2 / 4 / 8 at the 16 / 8 / 4-pair shards of an N = 2 / 4 / 8 run.  Measured (profiles/r06_input_sets.txt): a 16-pair shard
    # relaunched on ONE set of 1.4 GB runs 8 % faster than on two (the 256 MB memory-side cache keeps a sixth of it from launch
    # to launch); at 2.83 GB a second set changes 0.5 %.  Rounds 2-6 rotated only below 1 GB.
    nsets = args.input_sets if args.input_sets > 0 else max(1, -(-2800000000 // alg_bytes))
    sets = []
    for k in range(nsets):                      # set 0 is the one the oracle check and the CPU baseline look at
        t = synth.torch_inputs(device, B, C, H, W, fs=fs, flow_kind=args.flow, seed=plan["seed"] + 7919 * k)
        sets.append((t["x"], t["flow"], t["filt"], torch.zeros_like(t["x"])))   # caller-allocated, caller-zeroed output
    x, flow, filt, out = sets[0]
    steps, warmup, windows = cfg["steps"], cfg["warmup"], cfg["windows"]
    counter = [0]

    def step(ev=None):
        # one pass of the hot path over the batch; events (instrumented pass only) sit on the stream the kernel is launched on
        xi, fi, ki, oi = sets[counter[0] % nsets]
        counter[0] += 1
        if ev is not None:
            ev[0].record()
        err = my_lib.FilterInterpolationLayer_gpu_forward(xi, fi, ki, oi)
        if ev is not None:
            ev[1].record()
        if err != 0:
            raise RuntimeError("FilterInterpolationLayer_gpu_forward returned %d" % err)

    # Launch mode.  Rounds 2-5 replayed small shards from a HIP graph; measured in round 6 (profiles/r06_window_costs.txt):
    # the host needs ~10 us to enqueue a launch and a 4-frame shard runs 66 us -- eager launches keep the queue full, 300
    # eager steps take what 300 graph nodes take (65.2 / 64.8 us per step), and a 20-step window starts 35 us sooner (a graph
    # launch spends ~45 us between the stream reaching it and its first kernel).  Eager is the default for every shard size.
    use_graph = args.launch == "graph"
    for _ in range(args.prewarm):
        step()
    launch_stream = torch.cuda.current_stream(device)
    if use_graph:
        # the K steps of a window as ONE graph of K kernel nodes on a side stream, replayed once per window
        side = torch.cuda.Stream(device)
        graph = torch.cuda.CUDAGraph()
        torch.cuda.synchronize(device)
        with torch.cuda.graph(graph, stream=side):
            for _ in range(steps):
                step()
        for _ in range(warmup):
            step()
        with torch.cuda.stream(side):
            graph.replay()                      # one untimed replay (first replay uploads the graph)
        launch_stream = side

        def window(r):
            with torch.cuda.stream(side):
                graph.replay()
            return side
    else:
        for _ in range(warmup):
            step()

        def window(r):
            # K launches and NOTHING else between the two clock readings: no event is recorded inside the clock (rounds 1-5
            # put a pair around every launch: 6-10 us of stream time per step, BENCH_r05: 524.6 us per step around 514.2 us
            # kernels); the kernel's own duration comes from the instrumented passes below
            for _ in range(steps):
                step()
            return launch_stream
    worst_w, local_w = timed_windows(window, windows, world, device)
    worst, local = median(worst_w), median(local_w)
    # Outside the clock: (i) the same K steps bracketed by ONE HIP event pair on the launch stream -- the GPU-side span of a
    # window; (ii) eager: the same K launches with an event pair around EACH -- the dominant kernel's own duration
    spans = []
    for r in range(3):
        barrier_sync(world, device)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(launch_stream):
            e0.record()
        window(r)
        with torch.cuda.stream(launch_stream):
            e1.record()
        torch.cuda.synchronize(device)
        spans.append(e0.elapsed_time(e1) / 1e3)
    window_gpu_s = median(spans)
    per_launch_us = None
    if use_graph:
        avg_kernel_s = window_gpu_s / steps                 # (includes the ~1.5 us boundary between two dependent kernels)
    else:
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        barrier_sync(world, device)
        for e in ev:
            step(e)
        torch.cuda.synchronize(device)
        per_launch_us = [a.elapsed_time(b) * 1e3 for a, b in ev]
        avg_kernel_s = sum(per_launch_us) / steps / 1e6
    barrier_us = barrier_cost_us(world, device)
    achieved = alg_bytes / avg_kernel_s                                 # B/s, this rank's dominant kernel
    total_sites = plan["global_batch"] * H * W                          # all ranks' sites per step
    value = total_sites * steps / worst / 1e6
    dist_seen = None
    if _dist_on(world):
        # what the collective layer actually saw (evidence fields; nothing here is on the data path)
        import torch.distributed as dist
        gdev = device if dist.get_backend() == "nccl" else torch.device("cpu")      # gloo gathers host tensors only
        mine = torch.tensor([local / steps * 1e3, avg_kernel_s * 1e6, float(B), window_gpu_s / steps * 1e6],
                            dtype=torch.float64, device=gdev)
        gathered = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
        rows = [g.tolist() for g in gathered]
        dist_seen = {"backend": dist.get_backend(), "world_size_seen": dist.get_world_size(),
                     "single_rank_communicator": bool(_FORCE_DIST[0]),
                     "per_rank_ms_per_step": [round(r[0], 4) for r in rows],
                     "per_rank_avg_launch_us": [round(r[1], 2) for r in rows],
                     "per_rank_items": [int(r[2]) for r in rows],
                     "per_rank_window_gpu_us_per_step": [round(r[3], 2) for r in rows],
                     "barrier_us": barrier_us,
                     "worst_window_ms": [round(w * 1e3, 4) for w in worst_w],
                     "collectives": "1 broadcast of 8 int64 (configuration); per window 1 barrier BEFORE the clock starts and "
                                    "none before it stops (each rank stops its clock on its own synchronize); after the last "
                                    "window 1 max all-reduce of %d doubles (the window times), %d timed empty barriers "
                                    "(barrier_us), 1 all-gather of 4 doubles (this record)" % (windows, 9)}
        if dist.get_backend() == "nccl":
            try:
                dist_seen["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
            except Exception:                   # noqa: BLE001 -- version lookup is evidence only
                pass

    secondary = None
    if rank == 0 and world == 1 and not args.no_secondary:
        # the stress flow SURVEY.md section 8(d) asks for (i.i.d. N(0, 3^2) per pixel: defeats any tiling), same
        # launches / events, outside the timed region; and what a plain copy gets on this box
        t2 = synth.torch_inputs(device, B, C, H, W, fs=fs, flow_kind="iid", seed=plan["seed"] + 1)
        e = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(60)]
        for _ in range(20):
            my_lib.FilterInterpolationLayer_gpu_forward(t2["x"], t2["flow"], t2["filt"], out)
        for a0, a1 in e:
            a0.record()
            my_lib.FilterInterpolationLayer_gpu_forward(t2["x"], t2["flow"], t2["filt"], out)
            a1.record()
        torch.cuda.synchronize(device)
        iid_s = sum(a0.elapsed_time(a1) for a0, a1 in e) / len(e) / 1e3
        del t2
        secondary = {"iid_flow": {"avg_launch_us": round(iid_s * 1e6, 2), "mpixels_s": round(sites_per_launch / iid_s / 1e6, 1),
                                  "frac": round(alg_bytes / iid_s / HBM_PEAK_BPS, 4)},
                     "calibration": copy_calibration(my_lib, device, alg_bytes)}
        # the headline's own tensors in a layout a caller that owns its allocation could choose (NOT what the reference's callers hand
        # over, and not the headline): every row 64 floats longer, so that a tile's rows and a site's tap planes spread over all sixteen
        # 256-byte slots of a 4 KiB period instead of four (tools/synth.py: padded_planes; profiles/r06_plane_strides.txt)
        if (W * 4) % 1024 == 0:
            px, pf, pk, po = (synth.padded_planes(v) for v in (x, flow, filt, out))
            e = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(60)]
            for _ in range(20):
                my_lib.FilterInterpolationLayer_gpu_forward(px, pf, pk, po)
            for a0, a1 in e:
                a0.record()
                my_lib.FilterInterpolationLayer_gpu_forward(px, pf, pk, po)
                a1.record()
            torch.cuda.synchronize(device)
            pad_s = sum(a0.elapsed_time(a1) for a0, a1 in e) / len(e) / 1e3
            my_lib.FilterInterpolationLayer_gpu_forward(x, flow, filt, out)
            torch.cuda.synchronize(device)
            secondary["row_stride_padded_by_256B"] = {
                "avg_launch_us": round(pad_s * 1e6, 2), "mpixels_s": round(sites_per_launch / pad_s / 1e6, 1),
                "frac": round(alg_bytes / pad_s / HBM_PEAK_BPS, 4), "same_result": bool(torch.equal(po, out)),
                "note": "the headline call on views whose rows are 64 floats longer (a layout choice of the caller; the headline itself "
                        "runs on contiguous tensors, as the reference's callers provide them)"}
            del px, pf, pk, po
        if B == 32 and (C, H, W) == (3, 720, 1280):      # the default run: + the other BASELINE configs
            del sets[1:]
            try:
                secondary["rows"] = secondary_rows(my_lib, synth, torch, device, plan["seed"])
            except Exception as exc:            # noqa: BLE001 -- rows outside the timed region must not cost the headline its line
                secondary["rows"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
            try:
                secondary["rows"]["config4_memc_net_star_4x1280x720"] = config4_row(my_lib, torch, device, plan["seed"])
            except Exception as exc:            # noqa: BLE001
                secondary["rows"]["config4_memc_net_star_4x1280x720"] = {"error": "%s: %s" % (type(exc).__name__, exc)}

    if rank == 0:
        workload = "FilterInterpolation fwd fs=4 C=%d batch=%d %dx%d fp32 flow=%s" % (C, cfg["batch"], W, H, args.flow)
        traffic, traffic_source = load_traffic(workload) if B == cfg["batch"] else (None, None)   # (full batch only)
        line = {
            "metric": "Mpixels/s adaptive-warp fwd @720p batch32",
            "value": round(value, 1), "unit": "Mpixels/s", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": round(worst / steps * 1e3, 4), "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f32",
            "data": "synthetic" if not args.share_gpu else "synthetic; PLUMBING TEST (ranks share a GPU), not a measurement",
            "config": {"workload": workload, "batch_per_gpu": B, "global_batch": plan["global_batch"],
                       "prewarm_launches": args.prewarm, "launch": "hip_graph" if use_graph else "eager",
                       "windows": windows, "window": "median of %d windows of %d steps; clock: barrier + synchronize | t0 | "
                                                     "%d steps | this rank's synchronize | t1; max over ranks per window" % (
                                                         windows, steps, steps),
                       "window_ms_min_max": [round(min(worst_w) * 1e3, 4), round(max(worst_w) * 1e3, 4)],
                       "window_gpu_us_per_step": round(window_gpu_s / steps * 1e6, 2),
                       "window_host_enqueue_us": round(median(timed_windows.last_enqueue_s) * 1e6, 1),
                       "window_fixed_cost_us": round((worst - window_gpu_s) * 1e6, 1),   # wall - GPU span of the median window
                       "input_sets": nsets,
                       "sharding": "independent frame pairs, contiguous shards per rank, no data-path collective"},
            "roofline": {"bound": "hbm", "achieved": round(achieved / 1e9, 1), "peak": HBM_PEAK_BPS / 1e9,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_BPS, 4),
                         "traffic": traffic, "traffic_source": traffic_source,
                         "kernel": "fi_fwd", "algorithmic_bytes_per_launch": alg_bytes,
                         "avg_launch_us": round(avg_kernel_s * 1e6, 2),
                         "first_launches_us": [round(v, 2) for v in per_launch_us[:4]] if per_launch_us else None},
        }
        if secondary:
            # achievable_peak: what this box's HBM gives the tiled kernels' access pattern in this run (the better of a copy
            # and the kernel's own 7 : 1 read : write mix, memc_calibration_stream); the spec peak stays `peak`
            cal = secondary["calibration"]
            best = max(cal, key=cal.get)
            line["roofline"]["achievable_peak"] = round(cal[best] / 1e9, 1)
            line["roofline"]["achievable_peak_source"] = "memc_calibration_stream " + best
            line["roofline"]["calibration_GBps"] = {k: round(v / 1e9, 1) for k, v in cal.items()}
            line["roofline"]["frac_of_achievable"] = round(achieved / cal[best], 4)
            line["secondary"] = {"iid_flow": secondary["iid_flow"]}
            if "row_stride_padded_by_256B" in secondary:
                line["secondary"]["row_stride_padded_by_256B"] = secondary["row_stride_padded_by_256B"]
            line["secondary"].update(secondary.get("rows", {}))
        if dist_seen:
            line["dist"] = dist_seen
        failed = None
        if world == 1 and not args.no_cpu_baseline:
            # the timed launches' own output (input set 0 is relaunched so that `out` holds its result whatever the
            # rotation ended on) against the oracle result the baseline leg computes anyway
            my_lib.FilterInterpolationLayer_gpu_forward(x, flow, filt, out)
            torch.cuda.synchronize(device)
            line["cpu_baseline"], want = cpu_baseline(x, flow, filt, args.cpu_seconds)
            line["check"] = check_against_oracle(out, want)
            if not line["check"]["ok"]:
                failed = "output differs from the oracle by %.3g (> %g)" % (line["check"]["max_abs_err"], CHECK_TOLERANCE)
        print(json.dumps(line), flush=True)
        if failed:
            raise SystemExit("bench.py: " + failed)
    if _dist_on(world):
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
