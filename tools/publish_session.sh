#!/bin/bash
# tools/publish_session.sh <tag> [round] -- copy what a full GPU session (tools/gpu_session.sh <tag>) left under gpurun_out/<tag>/
# into profiles/ under the round's names (gpurun_out/ is scratch and untracked; profiles/ is what is committed and cited).
set -eu
TAG=$1; R=${2:-r05}
S=gpurun_out/$TAG; P=profiles
c() { [ -f "$S/$1" ] && cp "$S/$1" "$P/${R}_$2" || echo "missing: $S/$1"; }
c baselines.json baselines.json;            c baselines.log baselines.txt
c bench_kernel_stats.txt bench_kernel_stats.txt
grep '^{"metric"' "$S/bench.log" | tail -1 > "$P/${R}_bench_line.json"
c bench_line_profiled_run.json bench_line_profiled_run.json
c bench_model.json bench_model.json;        c bench_ops.json bench_ops.json;   c bench_ops.log bench_ops.txt
c bench_shards.log bench_shards.log;        c fi_bwd_nog1.txt fi_bwd_without_image_gradient.txt
c parity_errors.json parity_errors.json;    c traffic.json pmc_traffic.json;   c traffic_ops.json pmc_traffic_ops.json
c proj_ab.txt proj_ab.txt;                  c proj_burst.txt proj_burst_host_enqueue.txt
c proj_calls.txt proj_calls_kernel_trace.txt; c proj_owner5_phases.txt proj_owner5_phases.txt
c proj_owner5_sq.txt proj_owner5_sq_counters.txt; c pytest_gpu.log pytest_gpu.log; c sweep_kernel_stats.txt sweep_kernel_stats.txt
c proj_motion_sweep.txt proj_motion_sweep_after.txt; c motion_sweep_all.txt motion_sweep_all_operators.txt
cp "$S/traffic.json" "$P/traffic.json"      # what bench.py's roofline.traffic cites (checked against the kernel sources' hash)
echo "published $S -> $P/${R}_*"
