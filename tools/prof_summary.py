#!/usr/bin/env python
"""tools/prof_summary.py -- condense rocprofv3 output (the rocpd sqlite database it writes on ROCm 7.2) into the
small text/JSON summaries that are committed under profiles/.

    python tools/prof_summary.py stats   <results.db> [--out profiles/xxx_kernel_stats.txt]
    python tools/prof_summary.py pmc     <results.db> [--match fi_fwd] [--out profiles/xxx_pmc.json]

`stats`  : per kernel: calls, total / average / min / max duration (what `rocprofv3 --stats` tabulates).
`pmc`    : per kernel and counter: mean value per dispatch (one row per dispatch in the database).

gfx950 counter caveat (guides/MI355X_MICROARCH.md "HBM"): FETCH_SIZE counts 128-B read requests at 64 B, i.e.
reports exactly half of the bytes of a wide coalesced streaming read; both FETCH_SIZE and WRITE_SIZE are in
KiB.  The `pmc` summary therefore also prints fetch_bytes_corrected = 2 * 1024 * FETCH_SIZE and
write_bytes = 1024 * WRITE_SIZE; tools/pmc_traffic.py calibrates that correction on a plain copy of known size
in the same session before it is trusted.
"""
import argparse
import json
import re
import sqlite3
import sys


def short(name, n=70):
    name = re.sub(r"\(.*", "", name)                     # drop the argument list
    name = name.replace("void ", "")
    return name if len(name) <= n else name[:n - 3] + "..."


def stats(db, tail=0):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels "
        "group by name order by sum(duration) desc").fetchall()
    out = ["%-72s %7s %14s %12s %12s %12s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us")]
    for name, calls, tot, avg, mn, mx in rows:
        out.append("%-72s %7d %14.1f %12.2f %12.2f %12.2f" % (short(name), calls, tot / 1e3, avg / 1e3, mn / 1e3, mx / 1e3))
    if tail and rows:
        # the dominant kernel's LAST `tail` dispatches = the timed steps of bench.py (its untimed pre-warm and
        # warm-up launches come first and run at lower clocks, tools/timeline.py)
        top = rows[0][0]
        d = [r[0] for r in cur.execute("select duration from kernels where name = ? order by start", (top,)).fetchall()]
        d = d[-tail:]
        out.append("")
        out.append("%s: last %d dispatches (the timed steps): avg %.2f us, min %.2f, max %.2f" % (
            short(top), len(d), sum(d) / len(d) / 1e3, min(d) / 1e3, max(d) / 1e3))
    return "\n".join(out) + "\n"


def pmc(db, match):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute(
        "select kernel_name, counter_name, count(*), avg(value), avg(duration), avg(grid_size), avg(vgpr_count), "
        "avg(lds_block_size), avg(scratch_size) from counters_collection "
        "group by kernel_name, counter_name, grid_size").fetchall()
    res = []
    for kname, cname, n, val, dur, grid, vgpr, lds, scratch in rows:
        if match and match not in kname:
            continue
        rec = {"kernel": short(kname, 100), "counter": cname, "dispatches": n, "mean_value": val,
               "mean_duration_us": dur / 1e3, "grid": grid, "vgpr": vgpr, "lds_bytes": lds, "scratch": scratch}
        if cname == "FETCH_SIZE":
            rec["fetch_bytes_raw"] = val * 1024
            rec["fetch_bytes_corrected_x2"] = val * 2048
        if cname == "WRITE_SIZE":
            rec["write_bytes"] = val * 1024
        res.append(rec)
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("mode", choices=["stats", "pmc"])
    ap.add_argument("db")
    ap.add_argument("--match", default="")
    ap.add_argument("--out", default="")
    ap.add_argument("--tail", type=int, default=0, help="stats: also average the dominant kernel's last N dispatches")
    a = ap.parse_args()
    text = stats(a.db, a.tail) if a.mode == "stats" else json.dumps(pmc(a.db, a.match), indent=1) + "\n"
    if a.out:
        open(a.out, "w").write(text)
    sys.stdout.write(text)


if __name__ == "__main__":
    main()
