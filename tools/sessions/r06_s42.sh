#!/bin/bash
# Round 6, GPU session 42: the kernel trace of bench.py and the PMC traffic passes on the FINAL binary (the one session r06_s40 tested; r06c's were
# taken on a build whose product code was the same but whose template list was one argument shorter).
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r06_s42
mkdir -p "$OUT"
cd "$REPO"
sha256sum memc-net_amd/lib/libmemc_hip.so | tee "$OUT/lib_sha256.txt"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/prof_trace" -o bench -- python "$REPO/bench.py" --no-cpu-baseline --no-secondary > "$OUT/prof_trace.log" 2>&1
grep '^{"metric"' "$OUT/prof_trace.log" | tail -1 > "$OUT/bench_line_profiled_run.json"
python "$REPO/tools/prof_summary.py" stats "$OUT/prof_trace/bench_results.db" --tail 300 --out "$OUT/bench_kernel_stats.txt" | grep -v "^at::\|^$" | head -6
rm -rf "$OUT/prof_trace"
cd "$REPO" && timeout 900 python tools/pmc_traffic.py --out "$OUT" 2>&1 | tail -12
ls "$OUT"
