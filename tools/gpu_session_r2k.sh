#!/bin/bash
# fi_bwd_cn: per-kernel split of the many-channel FilterInterpolation backward
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT="$REPO/gpurun_out/r2k"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/prof" -o p -- python "$REPO/tools/bench_ops.py" --only fi_bwd_ctx --json "$OUT/bwd_ctx.json" > "$OUT/bwd_ctx.log" 2>&1
grep -v amdgpu.ids "$OUT/bwd_ctx.log" | tail -4
python "$REPO/tools/prof_summary.py" stats "$OUT/prof/p_results.db" --out "$OUT/kernel_stats.txt" | grep -v "^at::\|^$" | head -12
rm -rf "$OUT/prof"
