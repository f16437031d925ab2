#!/usr/bin/env python
"""tools/bench_model.py -- BASELINE.json config 4: full MEMC_Net_star inference (random weights) on 1280x720
frame pairs, sharded as independent pairs over the GPUs of one node.  NOT the headline metric (bench.py is);
this is the end-to-end context for it: how much of a frame interpolation is spent in the hot-path operators.

    python tools/bench_model.py [--pairs 4 --steps 5 --warmup 2]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        tools/bench_model.py --gpus N ...

Rank 0 builds the weights, all other ranks receive them through ONE bucketed RCCL broadcast
(networks/replicate.py); after that the ranks never communicate on the data path.  Frames are padded to
multiples of 128 by replication exactly like the reference demo (demo_HD720p.py:88-113) and the output is cropped
back.  Prints one JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "memc-net_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)


HOT_PATH_SUFFIXES = ("_gpu_forward", "_gpu_backward", "_gpu_forward_ws", "_gpu_backward_ws")


def hot_path_entry_points(my_lib):
    """Every operator entry point of the ctypes loader, the caller-workspace ones (`*_ws`: what the projection layers call
    since round 5 -- rounds 5's record missed both projections because this list ended at `_gpu_forward`) included."""
    return [n for n in dir(my_lib) if n.endswith(HOT_PATH_SUFFIXES) and callable(getattr(my_lib, n))]


def instrumented_pass(my_lib, torch, dev, fn):
    """One pass of `fn()` with a HIP event pair around every hot-path operator call (the my_package.functions.* layers look
    the entry points up on the my_lib module at call time) and one around the whole pass.  Returns
    (milliseconds per entry point, number of operator calls, milliseconds of the pass): event spans on the stream, not
    host timers."""
    spans, originals = [], {}
    for name in hot_path_entry_points(my_lib):
        f = originals[name] = getattr(my_lib, name)

        def wrapped(*args, _fn=f, _name=name):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            r = _fn(*args)
            e.record()
            spans.append((_name, s, e))
            return r
        setattr(my_lib, name, wrapped)
    try:
        s0, e0 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s0.record()
        fn()
        e0.record()
        torch.cuda.synchronize(dev)
    finally:
        for name, f in originals.items():
            setattr(my_lib, name, f)
    per_op = {}
    for name, s, e in spans:
        per_op[name] = per_op.get(name, 0.0) + s.elapsed_time(e)
    return per_op, len(spans), s0.elapsed_time(e0)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--pairs", type=int, default=4, help="frame pairs per GPU per step (32 / 8 GPUs)")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--height", type=int, default=720)
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--json", default="")
    ap.add_argument("--share-gpu", action="store_true", help=argparse.SUPPRESS)   # see bench.py
    ap.add_argument("--miopen-search", action="store_true",
                    help="torch.backends.cudnn.benchmark = True: let MIOpen time its convolution solvers per shape")
    ap.add_argument("--channels-last", action="store_true", help="NHWC activations / weights for the dense layers")
    a = ap.parse_args(argv)

    import torch
    import bench
    rank, local_rank, world = bench.dist_env()
    if world != a.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (a.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("needs a GPU: the HIP operators have no CPU fallback")
    dev_index = local_rank % torch.cuda.device_count() if a.share_gpu else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if a.share_gpu:                                # plumbing test on a box with fewer GPUs than ranks
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=dev)

    torch.backends.cudnn.benchmark = bool(a.miopen_search)
    import my_package._ext.my_lib as my_lib
    import networks
    torch.manual_seed(1234 + rank)                     # deliberately different: only the broadcast makes them equal
    net = networks.MEMC_Net_star(channel=3, filter_size=4, training=False).to(dev).eval()
    if a.channels_last:
        net = net.to(memory_format=torch.channels_last)
    t0 = time.perf_counter()
    msgs, nbytes = networks.broadcast_module_state(net, src=0)
    torch.cuda.synchronize(dev)
    bcast_s = time.perf_counter() - t0
    if world > 1:                                      # the replicas really are replicas: same checksum everywhere
        import torch.distributed as dist
        total = sum(float(v.double().abs().sum()) for v in net.state_dict().values())
        lo, hi = torch.tensor([total], dtype=torch.float64, device=dev), torch.tensor([total], dtype=torch.float64, device=dev)
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        if float(lo) != float(hi):
            raise SystemExit("weight broadcast left the ranks with different parameters")

    g = torch.Generator(device=dev).manual_seed(99 + rank)
    frames = torch.rand((2, a.pairs, 3, a.height, a.width), device=dev, generator=g)
    pl, pr, pt, pb = networks.pad_amounts(a.height, a.width)

    def interpolate_batch():
        return networks.interpolate_pairs(net, frames[0], frames[1])

    with torch.no_grad():
        worst, local = bench.timed_steps(lambda i: interpolate_batch(), a.steps, a.warmup, world, dev)
    dist_seen = None
    if world > 1:                                      # what the collective layer saw (evidence; not on the data path)
        import torch.distributed as dist
        gdev = dev if dist.get_backend() == "nccl" else torch.device("cpu")
        mine = torch.tensor([local / a.steps * 1e3, bcast_s], dtype=torch.float64, device=gdev)
        gathered = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
        dist_seen = {"backend": dist.get_backend(), "world_size_seen": dist.get_world_size(),
                     "per_rank_ms_per_step": [round(float(g[0]), 2) for g in gathered],
                     "weight_broadcast": {"messages": msgs, "bytes": int(nbytes),
                                          "per_rank_seconds": [round(float(g[1]), 4) for g in gathered],
                                          "GBps_slowest_rank": round(nbytes / max(float(g[1]) for g in gathered) / 1e9, 2)}}
    with torch.no_grad():
        # share of the step spent inside the hot-path operators: one extra untimed pass with events around them
        per_op, n_calls, pass_ms = instrumented_pass(my_lib, torch, dev, interpolate_batch)

    if rank == 0:
        pairs = world * a.pairs * a.steps
        line = {"metric": "MEMC_Net_star inference, interpolated %dx%d frames/s" % (a.width, a.height), "value": round(pairs / worst, 3),
                "unit": "frames/s", "mpixels_per_s": round(pairs * a.height * a.width / worst / 1e6, 2),
                "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(worst / a.steps * 1e3, 2),
                "scaling": "weak", "dtype": "f32", "data": "synthetic, random weights",
                "dense_layers": {"miopen_search": bool(a.miopen_search), "channels_last": bool(a.channels_last)},
                "config": {"workload": "MEMC_Net_star inference %dx%d (padded %dx%d), %d pairs/GPU" % (
                    a.width, a.height, a.width + pl + pr, a.height + pt + pb, a.pairs),
                    "global_pairs_per_step": world * a.pairs,
                    "weights": "rank 0 -> all, %d %s broadcast message(s), %.1f MB, %.3f s"
                               % (msgs, "gloo (PLUMBING TEST, ranks share a GPU)" if a.share_gpu else "RCCL",
                                  nbytes / 1e6, bcast_s)},
                "hot_path_ops_ms": {k: round(v, 3) for k, v in sorted(per_op.items())},
                "hot_path_ops_calls": n_calls,
                "hot_path_share_of_step": round(sum(per_op.values()) / pass_ms, 4), "instrumented_pass_ms": round(pass_ms, 2)}
        if dist_seen:
            line["dist"] = dist_seen
        print(json.dumps(line), flush=True)
        if a.json:
            json.dump(line, open(a.json, "w"), indent=1)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
