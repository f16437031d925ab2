#!/usr/bin/env python
"""tools/pmc_any.py -- HBM traffic (L2-miss bytes) per launch of ANY kernels of one `tools/bench_ops.py` invocation,
same method as tools/pmc_traffic.py: FETCH_SIZE / WRITE_SIZE in separate `rocprofv3 --pmc` passes, gfx950 factors
calibrated on a float4 copy of known size in the same session.  Run ON THE GPU BOX:

    python tools/pmc_any.py --out gpurun_out/<tag> --match fi_fwd_tiled -- --only fi_fwd --headline-only --variants=-1,15,16

Everything behind `--` goes to bench_ops.py.  Prints, per (kernel name, grid) group that matches, the mean read / written
bytes per dispatch and the mean profiled duration; writes <out>/pmc_any.json.
"""
import argparse
import json
import os
import sqlite3
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, ROOT)
from pmc_traffic import biggest, counter_rows, run_pmc     # noqa: E402


def main():
    argv = sys.argv[1:]
    extra = []
    if "--" in argv:
        i = argv.index("--")
        argv, extra = argv[:i], argv[i + 1:]
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "pmc_any"))
    ap.add_argument("--match", default="", help="kernel-name fragments, comma separated (default: every kernel of namespace memc)")
    ap.add_argument("--min-grid", type=int, default=0)
    a = ap.parse_args(argv)
    os.makedirs(a.out, exist_ok=True)
    py = sys.executable
    sweep = [py, os.path.join(ROOT, "tools", "bench_ops.py"), "--json", os.path.join(a.out, "ops_under_pmc.json")] + extra
    probe = [py, os.path.join(ROOT, "tools", "probes", "run_probe.py"), "copyonly"]
    dbs, cal = {}, {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        dbs[ctr] = run_pmc(ctr, a.out, "any", sweep)
        cal[ctr] = biggest(counter_rows(run_pmc(ctr, a.out, "copy", probe), "copy4"))
    known = (32 * 720 * 1280 * 96 // 32) * 16       # bytes the probe reads == bytes it writes
    k_fetch = known / (cal["FETCH_SIZE"][4] * 1024.0)
    k_write = known / (cal["WRITE_SIZE"][4] * 1024.0)
    frags = [f for f in a.match.split(",") if f] or ["memc::"]
    out = {"calibration": {"k_fetch": k_fetch, "k_write": k_write}, "bench_ops_args": extra, "kernels": []}
    seen = set()
    for frag in frags:
        fr = {(r[0], r[2]): r for r in counter_rows(dbs["FETCH_SIZE"], frag)}
        wr = {(r[0], r[2]): r for r in counter_rows(dbs["WRITE_SIZE"], frag)}
        for key in sorted(fr, key=lambda k: -k[1]):
            if key in seen or key not in wr or key[1] < a.min_grid:
                continue
            seen.add(key)
            f, w = fr[key], wr[key]
            rd, wb = k_fetch * f[4] * 1024, k_write * w[4] * 1024
            out["kernels"].append({"kernel": key[0].split("(")[0], "grid": key[1], "dispatches": f[3],
                                   "hbm_read_bytes": rd, "hbm_write_bytes": wb, "profiled_us": f[5] / 1e3})
            print("%-72s grid %9d  n %4d  read %8.1f MB  write %8.1f MB  %8.1f us" % (
                key[0].split("(")[0][:72], key[1], f[3], rd / 1e6, wb / 1e6, f[5] / 1e3))
    json.dump(out, open(os.path.join(a.out, "pmc_any.json"), "w"), indent=1)
    subprocess.run("rm -rf %s/pmc_any_* %s/pmc_copy_*" % (a.out, a.out), shell=True)


if __name__ == "__main__":
    main()
