"""bench.py's own line: the self-check against the oracle, the evidence fields, and -- where the box has two GPUs --
the real RCCL path (one process per GPU, backend "nccl").  GPU only."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, stdout[-2000:]
    return json.loads(lines[0])


def test_bench_line_checks_its_own_output_against_the_oracle():
    """A short run of the default workload's code path (2 frame pairs): the line carries `check` (max-abs error of the
    timed launches' output against the oracle, must be <= 1e-4), `roofline`, `cpu_baseline`, `traffic_source`, and the
    input sets it rotated over (a 2-frame launch moves 177 MB: eight sets)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--batch", "2", "--steps", "6", "--warmup", "2",
                        "--prewarm", "4", "--no-secondary", "--cpu-seconds", "0.5", "--launch", "eager"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:]
    line = _line(r.stdout)
    assert line["check"]["ok"] and line["check"]["max_abs_err"] <= 1e-4 and line["check"]["frames"] == 2
    assert line["config"]["input_sets"] >= 4
    assert line["roofline"]["bound"] == "hbm" and line["roofline"]["frac"] > 0
    assert "traffic_source" in line["roofline"]
    assert line["cpu_baseline"]["kind"] == "port" and line["cpu_baseline"]["cores"] >= 1
    assert line["n_gpus"] == 1 and line["dtype"] == "f32" and line["vs_baseline"] is None
    c = line["config"]
    assert c["windows"] == 7 and c["launch"] == "eager" and c["window_gpu_us_per_step"] > 0
    assert c["window_ms_min_max"][0] - 1e-3 <= line["ms_per_step"] * 6 <= c["window_ms_min_max"][1] + 1e-3
    assert len(line["roofline"]["first_launches_us"]) == 4


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs: the lease of this pool has one")
def test_two_ranks_over_rccl():
    """The N > 1 path as the driver launches it: one process per GPU, backend nccl (= RCCL), strong scaling.  The line
    must say what the collective layer saw."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29631", os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--steps", "20", "--warmup", "5", "--prewarm", "20"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-3000:]
    line = _line(r.stdout)
    d = line["dist"]
    assert d["backend"] == "nccl" and d["world_size_seen"] == 2
    assert d["per_rank_items"] == [16, 16] and len(d["per_rank_ms_per_step"]) == 2
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["config"]["global_batch"] == 32


def test_collective_layer_on_a_one_rank_rccl_communicator():
    """The lease has one GPU, so the N > 1 path has never met RCCL -- but a communicator of ONE rank is a real RCCL communicator:
    `--dist-single` sends the whole collective layer of the bench line through backend nccl (init with device_id, the int64
    broadcast, the barrier in front of every window, the float64 MAX all-reduce of the window times, the timed barriers, the
    all-gather), on device tensors, as the ranks of an 8-GPU job would."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dist-single", "--batch", "2", "--steps", "6",
                        "--warmup", "2", "--prewarm", "4", "--windows", "3", "--no-secondary", "--cpu-seconds", "0.5"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-3000:]
    line = _line(r.stdout)
    d = line["dist"]
    assert d["backend"] == "nccl" and d["world_size_seen"] == 1 and d["single_rank_communicator"] is True
    assert d["per_rank_items"] == [2] and len(d["worst_window_ms"]) == 3 and d["barrier_us"] > 0
    assert "rccl_version" in d
    assert line["n_gpus"] == 1 and line["check"]["ok"]                 # (still rank 0 at N = 1: the oracle check runs)


def test_two_ranks_sharing_one_gpu_plumbing():
    """The N > 1 code path end to end on a one-GPU box: two ranks on the same device, gloo standing in for RCCL (which
    refuses two ranks on one device).  Not a measurement -- the line says so in `data` -- but every collective, the shard
    plan, the HIP-graph replay of the rotating input sets and the `dist` record run for real."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29633", os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--share-gpu", "--batch", "8", "--steps", "12", "--warmup", "3", "--prewarm", "6",
                        "--launch", "graph", "--windows", "3"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-3000:]
    line = _line(r.stdout)
    d = line["dist"]
    assert d["backend"] == "gloo" and d["world_size_seen"] == 2 and d["per_rank_items"] == [4, 4]
    assert len(d["per_rank_ms_per_step"]) == 2 and all(t > 0 for t in d["per_rank_ms_per_step"])
    assert "PLUMBING TEST" in line["data"] and line["n_gpus"] == 2 and line["config"]["global_batch"] == 8
    assert line["config"]["launch"] == "hip_graph" and line["config"]["input_sets"] >= 4
    assert line["config"]["windows"] == 3 and len(d["worst_window_ms"]) == 3 and d["barrier_us"] > 0
    assert min(d["worst_window_ms"]) - 1e-3 <= line["ms_per_step"] * 12 <= max(d["worst_window_ms"]) + 1e-3      # the median window
    assert "cpu_baseline" not in line and "check" not in line          # rank 0 at N = 1 only
