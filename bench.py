#!/usr/bin/env python
"""bench.py -- headline benchmark of the adaptive-warp hot path on MI355X.

Metric (BASELINE.json): Mpixels/s of the fused adaptive-warp (FilterInterpolation) forward, 4x4 filter, C=3,
fp32, batch 32 of 1280x720 frames per GPU, synthetic inputs already resident in HBM.  One "step" = one pass of
the operator over one batch (one kernel launch through the C ABI, on torch's current stream).

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Multi-GPU: frame pairs are independent, so each rank owns its own batch-32 shard (weak scaling) and there is
no collective on the data path; RCCL is used only to broadcast the run configuration from rank 0 before the
timed region and to take the max of the per-rank times after it.

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
}


# ----------------------------------------------------------------------------------------------------------
# rank / sharding / timing plumbing (backend-agnostic so that it is testable with gloo on CPU)
# ----------------------------------------------------------------------------------------------------------
def dist_env():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def shard_plan(rank, world, batch_per_gpu, base_seed):
    """Weak scaling over independent frame pairs: every rank owns `batch_per_gpu` items of a global batch of
    world * batch_per_gpu; item g of the global batch lives on rank g // batch_per_gpu.  No halo, no exchange."""
    first = rank * batch_per_gpu
    return {"rank": rank, "world": world, "first_item": first, "items": batch_per_gpu,
            "global_batch": world * batch_per_gpu, "seed": base_seed + rank}


def broadcast_config(cfg, world, device):
    """Rank 0's run configuration wins (RCCL broadcast over xGMI on GPUs, gloo on CPU).  Ints only."""
    if world == 1:
        return cfg
    import torch
    import torch.distributed as dist
    keys = sorted(cfg)
    t = torch.tensor([int(cfg[k]) for k in keys], dtype=torch.int64, device=device)
    dist.broadcast(t, src=0)
    return {k: int(v) for k, v in zip(keys, t.tolist())}


def barrier_sync(world, device):
    import torch
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    if device.type == "cuda":
        torch.cuda.synchronize(device)


def timed_steps(step_fn, steps, warmup, world, device):
    """`warmup` untimed steps, then exactly `steps` steps bracketed by barrier + synchronize on both sides.
    Returns (max-over-ranks wall seconds, local wall seconds)."""
    import torch
    for _ in range(warmup):
        step_fn(None)
    barrier_sync(world, device)
    t0 = time.perf_counter()
    for i in range(steps):
        step_fn(i)
    barrier_sync(world, device)
    local = time.perf_counter() - t0
    worst = local
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([local], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        worst = float(t.item())
    return worst, local


# ----------------------------------------------------------------------------------------------------------
# CPU baseline (rank 0, N = 1 only): the oracle -- a port of the reference's CPU code -- on a bounded sample
# ----------------------------------------------------------------------------------------------------------
def cpu_baseline(x, flow, filt, target_seconds=12.0):
    """Times oracle.filter_interpolation_forward on the first few frames of the SAME tensors the GPU ran on."""
    import numpy as np
    from oracle import memc_oracle as O          # checker / baseline only -- never on the product path
    O.build()
    cores = O.num_threads()
    nb = min(x.shape[0], 4)
    xs, fs_, ks = (t[:nb].cpu().numpy() for t in (x, flow, filt))
    sites = nb * xs.shape[2] * xs.shape[3]
    O.filter_interpolation_forward(xs[:1], fs_[:1], ks[:1])     # page in
    reps, spent = 0, 0.0
    while spent < target_seconds and reps < 200:
        t0 = time.perf_counter()
        O.filter_interpolation_forward(xs, fs_, ks)
        spent += time.perf_counter() - t0
        reps += 1
    assert np.isfinite(spent)
    return {"value": round(sites * reps / spent / 1e6, 2), "unit": "Mpixels/s", "cores": cores, "kind": "port",
            "sample": "oracle FilterInterpolation fwd, first %d frames of the GPU batch (%dx%dx%dx%d), %d reps, "
                      "%.1f s, OpenMP over batch x rows" % (nb, nb, xs.shape[1], xs.shape[2], xs.shape[3], reps, spent)}


def load_traffic(workload_key):
    """HBM bytes per launch from the committed PMC profile of this same command (profiles/traffic.json,
    written by tools/pmc_traffic.py); None when no profile matches."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        data = json.load(open(path))
        rec = data.get(workload_key)
        return None if rec is None else rec.get("hbm_bytes_per_launch")
    except (OSError, ValueError):
        return None


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--prewarm", type=int, default=200,
                    help="untimed launches before the warm-up steps: the device needs ~100 launches (60 ms) to "
                         "reach its steady clocks (tools/timeline.py); never part of the timed region")
    ap.add_argument("--batch", type=int, default=32, help="frames per GPU")
    ap.add_argument("--height", type=int, default=720)
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--channels", type=int, default=3)
    ap.add_argument("--flow", default="smooth", choices=["smooth", "iid"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    # plumbing test of the N > 1 path on a box with fewer GPUs than ranks (ranks share devices, gloo instead of
    # RCCL, which refuses two ranks on one device): the line it prints is NOT a measurement
    ap.add_argument("--share-gpu", action="store_true", help=argparse.SUPPRESS)
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

    import my_package._ext.my_lib as my_lib                  # raises if libmemc_hip.so is missing
    from tools import synth

    cfg = broadcast_config({"batch": args.batch, "height": args.height, "width": args.width,
                            "channels": args.channels, "seed": 1234, "steps": args.steps,
                            "warmup": args.warmup}, world, device)
    B, C, H, W, fs = cfg["batch"], cfg["channels"], cfg["height"], cfg["width"], 4
    plan = shard_plan(rank, world, B, cfg["seed"])
    t = synth.torch_inputs(device, B, C, H, W, fs=fs, flow_kind=args.flow, seed=plan["seed"])
    x, flow, filt = t["x"], t["flow"], t["filt"]
    out = torch.zeros_like(x)                 # caller-allocated, caller-zeroed (reference contract)
    steps, warmup = cfg["steps"], cfg["warmup"]
    starts = [torch.cuda.Event(enable_timing=True) for _ in range(steps)]
    stops = [torch.cuda.Event(enable_timing=True) for _ in range(steps)]

    def step(i):
        # one pass of the hot path over the batch; events sit on the stream the kernel is launched on
        if i is not None:
            starts[i].record()
        err = my_lib.FilterInterpolationLayer_gpu_forward(x, flow, filt, out)
        if i is not None:
            stops[i].record()
        if err != 0:
            raise RuntimeError("FilterInterpolationLayer_gpu_forward returned %d" % err)

    for _ in range(args.prewarm):
        step(None)
    worst, _local = timed_steps(step, steps, warmup, world, device)
    sites_per_launch = B * H * W
    kernel_ms = [a.elapsed_time(b) for a, b in zip(starts, stops)]
    avg_kernel_s = sum(kernel_ms) / len(kernel_ms) / 1e3
    alg_bytes = BYTES_PER_SITE["fi_fwd"](C, fs) * sites_per_launch
    achieved = alg_bytes / avg_kernel_s                                 # B/s, this rank's dominant kernel
    value = world * sites_per_launch * steps / worst / 1e6

    if rank == 0:
        workload = "FilterInterpolation fwd fs=4 C=%d batch=%d %dx%d fp32 flow=%s" % (C, B, W, H, args.flow)
        traffic = load_traffic(workload)
        line = {
            "metric": "Mpixels/s adaptive-warp fwd @720p batch32",
            "value": round(value, 1), "unit": "Mpixels/s", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": round(worst / steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32",
            "data": "synthetic" if not args.share_gpu else "synthetic; PLUMBING TEST (ranks share a GPU), not a measurement",
            "config": {"workload": workload, "batch_per_gpu": B, "global_batch": plan["global_batch"],
                       "prewarm_launches": args.prewarm,
                       "sharding": "independent frame pairs per rank, no data-path collective"},
            "roofline": {"bound": "hbm", "achieved": round(achieved / 1e9, 1), "peak": HBM_PEAK_BPS / 1e9,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_BPS, 4),
                         "traffic": traffic,
                         "kernel": "fi_fwd", "algorithmic_bytes_per_launch": alg_bytes,
                         "avg_launch_us": round(avg_kernel_s * 1e6, 2)},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(x, flow, filt)
        print(json.dumps(line), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
