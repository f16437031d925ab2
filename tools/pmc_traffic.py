#!/usr/bin/env python
"""tools/pmc_traffic.py -- HBM traffic of the headline kernel from the rocprofv3 PMC counters, with the
gfx950 corrections calibrated in the same session.  Run ON THE GPU BOX:

    python tools/pmc_traffic.py --out gpurun_out/<tag>

Method (guides/MI355X_MICROARCH.md "HBM" / "rocprofv3 PMC slots"):
  * FETCH_SIZE and WRITE_SIZE cannot share a pass (TCC has 4 slots; they cost 3 + 2), so each is collected in
    its own `rocprofv3 --pmc <counter> --kernel-trace` run of the very same `bench.py` command;
  * both are in KiB; on gfx950 FETCH_SIZE tallies 128-B read requests at 64 B, i.e. reports HALF of the bytes
    of a wide coalesced read.  That factor is not assumed: a plain float4 copy of known size
    (tools/probes/io_skeleton.hip, probe_copy) is profiled in the same session and
        k_fetch = bytes_read_known / (FETCH_SIZE * 1024),   k_write = bytes_written_known / (WRITE_SIZE * 1024)
    are applied to the kernel's counters;
  * result per launch:  hbm_bytes = k_fetch * FETCH_SIZE * 1024 + k_write * WRITE_SIZE * 1024.

Writes <out>/traffic.json (copy it to profiles/traffic.json, which bench.py reads) and prints a summary.
"""
import argparse
import json
import os
import sqlite3
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import kernel_source_hash  # noqa: E402


def run_pmc(counter, outdir, name, cmd):
    d = os.path.join(outdir, "pmc_%s_%s" % (name, counter))
    env = dict(os.environ, TMPDIR="/tmp")
    subprocess.run(["rocprofv3", "--pmc", counter, "--kernel-trace", "-d", d, "-o", "r", "--"] + cmd,
                   cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, check=True, timeout=900)
    return os.path.join(d, "r_results.db")


def counter_rows(db, match):
    cur = sqlite3.connect(db).cursor()
    return cur.execute(
        "select kernel_name, counter_name, grid_size, count(*), avg(value), avg(duration) from counters_collection "
        "where kernel_name like ? group by kernel_name, counter_name, grid_size", ("%" + match + "%",)).fetchall()


def biggest(rows):
    """the (kernel, grid) group with the largest grid: the full-size launches (warm-ups share it)"""
    return max(rows, key=lambda r: r[2]) if rows else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "pmc"))
    ap.add_argument("--kernel", default="fi_fwd_tiled")
    ap.add_argument("--bench-args", default="--steps 6 --warmup 2 --no-cpu-baseline --no-secondary --launch eager")
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    py = sys.executable
    bench = [py, os.path.join(ROOT, "bench.py")] + a.bench_args.split()
    probe = [py, os.path.join(ROOT, "tools", "probes", "run_probe.py"), "copyonly"]
    probe_so = os.path.join(ROOT, "tools", "probes", "libio_skeleton.so")
    if not os.path.exists(probe_so):            # (built here, not under the profiler: a fresh checkout has no probe library)
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-o", probe_so,
                        os.path.join(ROOT, "tools", "probes", "io_skeleton.hip")], check=True)

    res = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        k = biggest(counter_rows(run_pmc(ctr, a.out, "bench", bench), a.kernel))
        c = biggest(counter_rows(run_pmc(ctr, a.out, "copy", probe), "copy4"))
        res[ctr] = {"kernel": k, "copy": c}

    n4 = 32 * 720 * 1280 * 96 // 32                 # float4 elements the probe copies (run_probe.py copyonly)
    known = n4 * 16                                 # bytes read == bytes written
    k_fetch = known / (res["FETCH_SIZE"]["copy"][4] * 1024.0)
    k_write = known / (res["WRITE_SIZE"]["copy"][4] * 1024.0)
    fetch_kib, write_kib = res["FETCH_SIZE"]["kernel"][4], res["WRITE_SIZE"]["kernel"][4]
    hbm = k_fetch * fetch_kib * 1024 + k_write * write_kib * 1024
    line = subprocess.run(bench, stdout=subprocess.PIPE, text=True, check=True).stdout.strip().splitlines()[-1]
    workload = json.loads(line)["config"]["workload"]
    alg = json.loads(line)["roofline"]["algorithmic_bytes_per_launch"]
    rec = {
        "kernel": res["FETCH_SIZE"]["kernel"][0].split("(")[0],
        "dispatches_averaged": res["FETCH_SIZE"]["kernel"][3],
        "FETCH_SIZE_KiB": fetch_kib, "WRITE_SIZE_KiB": write_kib,
        "calibration": {"probe": "copy4 (float4 grid-stride copy), %d bytes each way" % known,
                        "copy_FETCH_SIZE_KiB": res["FETCH_SIZE"]["copy"][4],
                        "copy_WRITE_SIZE_KiB": res["WRITE_SIZE"]["copy"][4],
                        "k_fetch": k_fetch, "k_write": k_write},
        "hbm_bytes_per_launch": hbm, "algorithmic_bytes_per_launch": alg, "traffic_over_algorithmic": hbm / alg,
        "profiled_launch_us": res["FETCH_SIZE"]["kernel"][5] / 1e3,
        # bench.py quotes this record only while the kernel sources are the ones it was measured on
        "kernel_source_hash": kernel_source_hash(),
    }
    out = {workload: rec}
    json.dump(out, open(os.path.join(a.out, "traffic.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))
    # the raw databases are large; the json above is what gets committed
    subprocess.run("rm -rf %s/pmc_bench_* %s/pmc_copy_*" % (a.out, a.out), shell=True)


if __name__ == "__main__":
    main()
