#!/bin/bash
# One GPU-box session: smoke, parity tests (with the record of observed errors), operator sweep, baselines, headline bench,
# shard sizes, model bench, rocprofv3 kernel traces and the PMC passes (FETCH_SIZE / WRITE_SIZE need separate passes: TCC has
# 4 slots, they cost 3 + 2).
# Usage (from the repo root, via gpurun):  bash tools/gpu_session.sh <tag> [quick]
set -u
TAG=${1:-r03}
QUICK=${2:-}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$REPO"
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== smoke";   timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee "$OUT/smoke.log"
echo "== pytest";  timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -15 | tee "$OUT/pytest_gpu.log"
cp gpurun_out/parity_errors.json "$OUT/parity_errors.json" 2>/dev/null
echo "== sweep (product library)";   timeout 900 python tools/bench_ops.py ${QUICK:+--quick} --json "$OUT/bench_ops.json" 2>&1 | grep -v amdgpu.ids | tee "$OUT/bench_ops.log"
echo "== baselines (reference kernels on this GPU, CPU oracle)"; timeout 900 python tests/bench_baselines.py --json "$OUT/baselines.json" 2>&1 | grep -v amdgpu.ids | tee "$OUT/baselines.log"
echo "== bench";   timeout 900 python bench.py 2>&1 | tail -3 | tee "$OUT/bench.log"
echo "== bench, strong-scaling shard sizes on one GPU (what rank 0 of an N-GPU run executes: batch 32 / N, HIP graph, rotating input sets)"
for b in 16 8 4; do timeout 300 python bench.py --batch $b --no-cpu-baseline --no-secondary 2>&1 | tail -1 | tee -a "$OUT/bench_shards.log"; done
echo "== model"; timeout 600 python tools/bench_model.py --json "$OUT/bench_model.json" 2>&1 | tail -2 | tee "$OUT/bench_model.log"
echo "== RGB backward: phase clocks of the product kernel and of the rounds 1-2 kernel (measurement build)"
for k in fi_bwd_pk2 fi_bwd; do timeout 300 python tools/trace_kernel.py $k 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/fi_bwd_traces.txt"; done
echo "== RGB backward A/B in one process: product against the rounds 1-2 kernel (arm 0)"
timeout 600 python tools/bench_ops.py --only fi_bwd --bwd-variants 0 --json "$OUT/bench_fi_bwd_ab.json" 2>&1 | grep -v amdgpu.ids | tee "$OUT/bench_fi_bwd_ab.log"
echo "== race screen"; timeout 600 python tools/stress_projection.py 60 2>&1 | tail -3 | tee "$OUT/stress.log"
cd /tmp && export TMPDIR=/tmp
echo "== rocprof kernel trace of bench.py (the same command as the bench line above, minus the CPU baseline and the secondary rows)"
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/prof_trace" -o bench -- python "$REPO/bench.py" --no-cpu-baseline --no-secondary > "$OUT/prof_trace.log" 2>&1
grep '^{"metric"' "$OUT/prof_trace.log" | tail -1 > "$OUT/bench_line_profiled_run.json"    # the bench line of THIS process
python "$REPO/tools/prof_summary.py" stats "$OUT/prof_trace/bench_results.db" --tail 300 --out "$OUT/bench_kernel_stats.txt" | grep -v "^at::\|^$" | head -6
rm -rf "$OUT/prof_trace"
echo "== rocprof kernel trace of the operator sweep"
timeout 900 rocprofv3 --kernel-trace --stats -d "$OUT/prof_sweep" -o sweep -- python "$REPO/tools/bench_ops.py" --quick --json "$OUT/bench_ops_profiled.json" > "$OUT/prof_sweep.log" 2>&1
python "$REPO/tools/prof_summary.py" stats "$OUT/prof_sweep/sweep_results.db" --out "$OUT/sweep_kernel_stats.txt" | grep -v "^at::\|^$" | head -40
rm -rf "$OUT/prof_sweep"
echo "== PMC traffic (FETCH_SIZE / WRITE_SIZE in separate passes, calibrated on a known copy)"
cd "$REPO" && timeout 900 python tools/pmc_traffic.py --out "$OUT" 2>&1 | tail -30
echo "== PMC traffic of the other kernels"
timeout 1500 python tools/pmc_traffic_ops.py --out "$OUT" 2>&1 | tail -14
ls "$OUT"
