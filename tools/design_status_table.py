#!/usr/bin/env python
"""tools/design_status_table.py <bench_line.json> -- the status table of DESIGN.md section 0 from a bench line: one markdown row per
row of the line (headline + secondary), microseconds per launch, fraction of the 8 TB/s specification, of the box's own copy
rate (roofline.achievable_peak), and how it was timed."""
import json
import sys

SECTION = {"iid_flow": "a1", "config2_fi_fwd": "a1", "config2_fi_bwd": "a2", "fi_bwd": "a2", "config3_flow_projection_fwd": "a3",
           "config3_depth_flow_projection_fwd": "a5", "flow_projection_fwd": "a3", "config3_flow_projection_bwd": "a4",
           "config3_depth_flow_projection_bwd": "a5", "interpolation": "a6", "config5": "a1", "context_warp": "a1", "config4": "f-1", "row_stride": "a1"}


def main():
    d = json.load(open(sys.argv[1]))
    r = d["roofline"]
    copy = r.get("achievable_peak")
    print("| row of the bench line (`%s`) | §8 row | µs per launch | of 8 TB/s | of the box's copy rate (%s GB/s) | timing |" % (sys.argv[1], copy))
    print("|---|---|---|---|---|---|")
    print("| **headline: FilterInterpolation fwd C=3, 32×720×1280** — %.0f Mpixels/s | a1 | %.1f | **%.1f %%** | %s | event pair per launch |" % (
        d["value"], r["avg_launch_us"], 100 * r["frac"], ("%.1f %%" % (100 * r["frac_of_achievable"])) if "frac_of_achievable" in r else ""))
    for k, v in d.get("secondary", {}).items():
        if not isinstance(v, dict):
            continue
        sec = next((s for p, s in SECTION.items() if k.startswith(p)), "")
        if "frac" in v:
            extra = ""
            if "cache_warm_us" in v:
                extra = " (cache-warm %.1f)" % v["cache_warm_us"]
            if "layer_call_us" in v:
                extra += " (through the Python layer: %.1f)" % v["layer_call_us"]
            of_copy = ("%.1f %%" % (100 * v["frac"] * 8000.0 / copy)) if copy else ""
            print("| `%s` | %s | %.1f%s | %.1f %% | %s | %s |" % (k, sec, v["avg_launch_us"], extra, 100 * v["frac"], of_copy, v.get("timing", "single")))
        elif "frames_per_s" in v:
            print("| `%s` | %s | %.1f ms per step = %.1f frames/s; hot path %.2f ms in %d calls = %.2f %% of the step; set-up %.2f s (%s) | — | — | %d warm-up + %d timed steps |" % (
                k, sec, v["ms_per_step"], v["frames_per_s"], v["hot_path_ms"], v["hot_path_calls"], 100 * v["hot_path_share"], v["setup_s"],
                v.get("miopen_cache"), v["warmup"], v["steps"]))
    cb, ck = d.get("cpu_baseline"), d.get("check")
    print()
    print("cpu_baseline:", cb)
    print("check:", ck)
    print("traffic:", r.get("traffic"), r.get("traffic_source"))


if __name__ == "__main__":
    main()
