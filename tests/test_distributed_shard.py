"""The N > 1 plumbing of bench.py (sharding plan, config broadcast, barrier-bracketed timing, max over
ranks) exercised with world_size 2 on the gloo backend.  CPU only: the step function here is a stand-in that
sleeps -- it checks the orchestration, not the kernels."""
import os
import socket
import sys

import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import time
    import torch
    import torch.distributed as dist
    import bench
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cpu")
    assert bench.dist_env() == (rank, rank, world)
    # rank 0's configuration wins
    mine = {"batch": 32 if rank == 0 else 7, "height": 720, "width": 1280, "seed": 1234 + 100 * rank}
    cfg = bench.broadcast_config(mine, world, dev)
    plan = bench.shard_plan(rank, world, cfg["batch"], cfg["seed"])
    calls = []

    def step(i):
        calls.append(i)
        time.sleep(0.02 if rank == 0 else 0.05)          # rank 1 is the slow one
    worst, local = bench.timed_steps(step, steps=4, warmup=2, world=world, device=dev)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, cfg, plan, calls, worst, local))


def test_two_rank_orchestration():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, cfg0, plan0, calls0, worst0, local0), (r1, cfg1, plan1, calls1, worst1, local1) = res
    assert cfg0 == cfg1 == {"batch": 32, "height": 720, "width": 1280, "seed": 1234}
    # weak scaling over independent frame pairs: disjoint, contiguous, complete
    assert (plan0["first_item"], plan0["items"]) == (0, 32) and (plan1["first_item"], plan1["items"]) == (32, 32)
    assert plan0["global_batch"] == plan1["global_batch"] == 64
    assert plan0["seed"] != plan1["seed"]
    # warmup steps are untimed (None), then exactly K timed steps
    assert calls0 == calls1 == [None, None, 0, 1, 2, 3]
    # every rank reports the MAX over ranks, i.e. the slow rank's time
    assert abs(worst0 - worst1) < 1e-9
    assert worst0 >= max(local0, local1) - 1e-9 and worst0 >= 4 * 0.05


def _window_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import time
    import torch
    import torch.distributed as dist
    import bench
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cpu")
    if rank == 1:
        # rank 1 is late into EVERY collective by 200 ms: a collective inside the clock would put those 200 ms into
        # rank 0's window (it waits for rank 1 there)
        real_barrier, real_all_reduce = dist.barrier, dist.all_reduce

        def late_barrier(*a, **k):
            time.sleep(0.2)
            return real_barrier(*a, **k)

        def late_all_reduce(*a, **k):
            time.sleep(0.2)
            return real_all_reduce(*a, **k)
        dist.barrier, dist.all_reduce = late_barrier, late_all_reduce
    order = []

    def window(r):
        order.append(("window", r))
        time.sleep(0.010 if rank == 0 else 0.020)        # the window's own work; rank 1 is the slow one
        return None
    worst, local = bench.timed_windows(window, 5, world, dev)
    cost = bench.barrier_cost_us(world, dev, reps=3)
    dist.destroy_process_group()
    q.put((rank, worst, local, order, cost))


def test_closing_collective_is_outside_the_timed_window():
    """a window ends on the rank's own synchronize; the max over ranks is taken afterwards.  Rank 1
    enters every collective 200 ms late -- none of that may appear in any window time, on either rank (the bounds leave 100 ms
    for a loaded host's scheduling hiccups: the windows' own work is 10 / 20 ms)."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_window_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, worst0, local0, order0, cost0), (_, worst1, local1, order1, cost1) = res
    assert order0 == order1 == [("window", r) for r in range(5)]
    assert worst0 == worst1 and len(worst0) == 5                       # one max-over-ranks time per window, same on both
    for w, l0, l1 in zip(worst0, local0, local1):
        assert w == max(l0, l1)
        assert 0.020 <= w < 0.120, (w, "a 200 ms collective delay leaked into the window")
        assert 0.010 <= l0 < 0.110                                       # rank 0 never waited for rank 1 inside its clock
    assert sorted(worst0)[2] < 0.120
    assert cost0 is not None and cost0 > 0                             # barrier_us: evidence field, outside the windows


def test_median_and_windowed_steps_single_rank():
    sys.path.insert(0, ROOT)
    import torch
    import bench
    assert bench.median([3.0, 1.0, 2.0]) == 2.0 and bench.median([4.0, 1.0, 2.0, 3.0]) == 2.5
    calls = []
    worst, local = bench.timed_steps(calls.append, steps=3, warmup=2, world=1, device=torch.device("cpu"), windows=4)
    assert calls == [None, None] + [0, 1, 2] * 4 and worst == local and worst >= 0
    assert bench.barrier_cost_us(1, torch.device("cpu")) is None


def test_single_rank_plan():
    sys.path.insert(0, ROOT)
    import bench
    plan = bench.shard_plan(0, 1, 32, 1234)
    assert plan["global_batch"] == 32 and plan["items"] == 32 and plan["seed"] == 1234
    assert bench.BYTES_PER_SITE["fi_fwd"](3, 4) == 96 and bench.BYTES_PER_SITE["fi_fwd"](64, 4) == 584


def test_strong_scaling_plan():
    """the global batch stays 32 frame pairs, rank r owns a contiguous 32 / N of them."""
    sys.path.insert(0, ROOT)
    import bench
    for world in (1, 2, 4, 8, 3, 5):
        plans = [bench.shard_plan(r, world, 32, 1234, "strong") for r in range(world)]
        assert all(p["global_batch"] == 32 and p["scaling"] == "strong" for p in plans)
        assert sum(p["items"] for p in plans) == 32
        assert plans[0]["first_item"] == 0
        for a, b in zip(plans, plans[1:]):
            assert b["first_item"] == a["first_item"] + a["items"]            # contiguous, disjoint, complete
        assert max(p["items"] for p in plans) - min(p["items"] for p in plans) <= 1
        assert len({p["seed"] for p in plans}) == world
    assert [bench.shard_plan(r, 8, 32, 0, "strong")["items"] for r in range(8)] == [4] * 8
    assert [bench.shard_plan(r, 3, 32, 0, "strong")["items"] for r in range(3)] == [11, 11, 10]
    # weak: every rank its own batch
    assert [bench.shard_plan(r, 4, 32, 0, "weak")["first_item"] for r in range(4)] == [0, 32, 64, 96]


def _strong_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import bench
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cpu")
    cfg = bench.broadcast_config({"batch": 32 if rank == 0 else 5, "seed": 1234 + rank}, world, dev)
    plan = bench.shard_plan(rank, world, cfg["batch"], cfg["seed"], "strong")
    # every rank processes ITS items; together they cover the global batch exactly once
    mine = torch.zeros(cfg["batch"], dtype=torch.int64)
    mine[plan["first_item"]:plan["first_item"] + plan["items"]] += 1
    dist.all_reduce(mine)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, plan, mine.tolist()))


def test_strong_scaling_two_ranks():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_strong_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, plan0, cover0), (_, plan1, cover1) = res
    assert (plan0["first_item"], plan0["items"]) == (0, 16) and (plan1["first_item"], plan1["items"]) == (16, 16)
    assert plan0["global_batch"] == plan1["global_batch"] == 32
    assert cover0 == cover1 == [1] * 32


def _bcast_worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, "memc-net_amd"))
    import importlib.util
    import torch
    import torch.distributed as dist
    # load replicate.py on its own: `import networks` would pull in my_package (needs the HIP library + a GPU)
    spec = importlib.util.spec_from_file_location(
        "replicate", os.path.join(ROOT, "memc-net_amd", "networks", "replicate.py"))
    rep = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rep)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(100 + rank)                                   # ranks start with DIFFERENT weights
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3), torch.nn.BatchNorm2d(8), torch.nn.Conv2d(8, 4, 1))
    before = torch.cat([t.double().reshape(-1) for t in net.state_dict().values()]).sum().item()
    msgs, nbytes = rep.broadcast_module_state(net, src=0, bucket_bytes=512)   # tiny buckets: several messages
    after = torch.cat([t.double().reshape(-1) for t in net.state_dict().values()]).sum().item()
    live = sum(p.double().sum().item() for p in net.parameters())   # the LIVE parameters changed, not copies
    pairs = list(rep.shard_pairs(rank, world, 4))
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, before, after, live, msgs, nbytes, pairs))


def test_weight_broadcast_two_ranks():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bcast_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, b0, a0, l0, m0, n0, p0), (_, b1, a1, l1, m1, n1, p1) = res
    assert b0 != b1                       # different before
    assert a0 == a1 == b0                 # rank 0's weights everywhere after (bit-exact: it is a copy)
    assert l0 == l1
    assert m0 == m1 and m0 >= 3 and n0 == n1          # fp32 buckets of <= 512 B + one int64 (num_batches_tracked)
    assert p0 == [0, 1, 2, 3] and p1 == [4, 5, 6, 7]  # disjoint, contiguous, complete
