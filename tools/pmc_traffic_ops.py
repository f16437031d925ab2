#!/usr/bin/env python
"""tools/pmc_traffic_ops.py -- HBM traffic of the OTHER kernels of the path (tools/pmc_traffic.py does the
headline one), same method: FETCH_SIZE / WRITE_SIZE in separate `rocprofv3 --pmc` passes over one operator sweep
(`tools/bench_ops.py`, 720p batch 32), gfx950 factors calibrated on a copy of known size in the same session.
Run ON THE GPU BOX:   python tools/pmc_traffic_ops.py --out gpurun_out/<tag>
Writes <out>/traffic_ops.json: per kernel the HBM bytes per launch next to its algorithmic bytes.
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from pmc_traffic import biggest, counter_rows, run_pmc     # noqa: E402

SITES = 32 * 720 * 1280
# kernel-name fragment -> (algorithmic bytes per site, sites per launch); every tensor touched once
KERNELS = {
    "fi_bwd_c3_pk": (180, SITES),                     # x 12 + flow 8 + taps 64 + gout 12 | gin1 12 + gin2 8 + gin3 64
    "fi_fwd_tiled_c4n": (4 * (2 * 64 + 2 + 16), SITES // 4),
    "fi_fwd_blend_c3": (188, SITES),
    "proj_owner5<false": (20, SITES),                 # flow 8 | count 4 + out 8
    "proj_owner5<true": (24, SITES),
    "proj_fill_pending": (0.5, SITES),                # (booked like round 3's proj_fillhole_carry: the holes and what their walks read)
    "proj_bwd_tiled<false": (28, SITES),              # flow 8 + count 4 + gout 8 | gin 8
    "proj_bwd_tiled<true": (44, SITES),               # flow 8 + depth 4 + count 4 + out 8 + gout 8 | gin1 8 + gin2 4
    "bl_fwd_tiled<3": (32, SITES),                    # x 12 + flow 8 | out 12
    "bl_bwd_c3_pk": (52, SITES),                      # x 12 + flow 8 + gout 12 | gin1 12 + gin2 8
    # FilterInterpolation backward, C = 64, batch 8 (fi_bwd_cn.hip): the operator's 912 B/site split over its kernels
    "fi_bwd_taps_c4n": (4 * (2 * 64 + 2 + 16 + 2 + 16), SITES // 4),     # x, gout, flow, taps | gin2, gin3
    "fi_bwd_image_owner": (4 * (64 + 64 + 2 + 16), SITES // 4),          # gout, flow, taps | gin1
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "pmc_ops"))
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    py = sys.executable
    sweep = [py, os.path.join(ROOT, "tools", "bench_ops.py"), "--only", "fi_fwd,fi_blend,fi_bwd,fi_bwd_ctx,proj,interp",
             "--ctx-only", "--variants=-1", "--json", os.path.join(a.out, "sweep_under_pmc.json")]
    probe = [py, os.path.join(ROOT, "tools", "probes", "run_probe.py"), "copyonly"]
    dbs, cal = {}, {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        dbs[ctr] = run_pmc(ctr, a.out, "ops", sweep)
        cal[ctr] = biggest(counter_rows(run_pmc(ctr, a.out, "copy", probe), "copy4"))
    known = (32 * 720 * 1280 * 96 // 32) * 16       # bytes the probe reads == bytes it writes
    k_fetch = known / (cal["FETCH_SIZE"][4] * 1024.0)
    k_write = known / (cal["WRITE_SIZE"][4] * 1024.0)
    out = {"calibration": {"k_fetch": k_fetch, "k_write": k_write}, "kernels": {}}
    for frag, (bps, sites) in KERNELS.items():
        f = biggest(counter_rows(dbs["FETCH_SIZE"], frag))
        w = biggest(counter_rows(dbs["WRITE_SIZE"], frag))
        if not f or not w:
            out["kernels"][frag] = None
            continue
        rd, wr = k_fetch * f[4] * 1024, k_write * w[4] * 1024
        alg = bps * sites
        out["kernels"][frag] = {"dispatches_averaged": f[3], "grid": f[2], "hbm_read_bytes": rd,
                                "hbm_write_bytes": wr, "hbm_bytes_per_launch": rd + wr,
                                "algorithmic_bytes_per_launch": alg,
                                "traffic_over_algorithmic": (rd + wr) / alg}
    json.dump(out, open(os.path.join(a.out, "traffic_ops.json"), "w"), indent=1)
    for k, v in out["kernels"].items():
        print("%-34s %s" % (k, "not seen" if v is None else "read %.0f MB  write %.0f MB  = %.2fx algorithmic (%d launches)" % (
            v["hbm_read_bytes"] / 1e6, v["hbm_write_bytes"] / 1e6, v["traffic_over_algorithmic"], v["dispatches_averaged"])))
    subprocess.run("rm -rf %s/pmc_ops_* %s/pmc_copy_*" % (a.out, a.out), shell=True)


if __name__ == "__main__":
    main()
