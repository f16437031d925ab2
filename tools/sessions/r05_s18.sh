#!/bin/bash
# Round 5, GPU session 18 (after the full session r05b): FilterInterpolation forward with four channels or more at ragged widths
# on the chunk pipeline -- the tests of that operator and of the kernel paths, then what depends on the sources' hash again:
# the headline's PMC traffic, the bench line, and the kernel trace of the same command.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05_s18
mkdir -p "$OUT"
cd "$REPO"
timeout 1500 python -m pytest tests -q -m gpu -k "ragged or paths or unaligned or multiples or fi_ or FilterInterpolation or filter_interpolation or forward or context or network or config or golden or reference" 2>&1 | tail -4 | tee $OUT/pytest.log
timeout 300 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee $OUT/fi_fwd_channels_ragged_width.txt
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "memc-net_amd"))
import torch
import my_package._ext.my_lib as L
from tools import synth
dev = torch.device("cuda:0")
print("FilterInterpolation forward, 8 x C x 720 x W, smooth flow; us per call (median of 9)")
for C in (8, 64):
    row = []
    for W in (1280, 1278):
        t = synth.torch_inputs(dev, 8, C, 720, W, flow_kind="smooth")
        x, f, k = t["x"], t["flow"], t["filt"]
        out = torch.zeros_like(x)
        for _ in range(3):
            L.FilterInterpolationLayer_gpu_forward(x, f, k, out)
        ts = []
        for _ in range(9):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); L.FilterInterpolationLayer_gpu_forward(x, f, k, out); b.record(); b.synchronize()
            ts.append(a.elapsed_time(b) * 1e3)
        ts.sort()
        row.append((W, ts[4], L.last_kernel_path()))
    print("C = %2d: " % C + "   ".join("W = %d: %8.1f us (%s)" % r for r in row) + "   ratio %.2f" % (row[1][1] / row[0][1]))
PY
timeout 900 python tools/pmc_traffic.py --out "$OUT" 2>&1 | tail -12
cp "$OUT/traffic.json" profiles/traffic.json
timeout 900 python bench.py 2>&1 | tail -1 > "$OUT/bench.log"; cut -c1-400 "$OUT/bench.log"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/prof_trace" -o bench -- python "$REPO/bench.py" --no-cpu-baseline --no-secondary > "$OUT/prof_trace.log" 2>&1
grep '^{"metric"' "$OUT/prof_trace.log" | tail -1 > "$OUT/bench_line_profiled_run.json"
python "$REPO/tools/prof_summary.py" stats "$OUT/prof_trace/bench_results.db" --tail 300 --out "$OUT/bench_kernel_stats.txt" | grep -v "^at::\|^$" | head -6
rm -rf "$OUT/prof_trace"
