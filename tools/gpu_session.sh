#!/bin/bash
# One GPU-box session: smoke, parity tests, operator sweep, headline bench, rocprofv3 kernel trace and the
# two PMC passes (FETCH_SIZE / WRITE_SIZE need separate passes: TCC has 4 slots, they cost 3 + 2).
# Usage (from the repo root, via gpurun):  bash tools/gpu_session.sh <tag> [quick]
set -u
TAG=${1:-r01}
QUICK=${2:-}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$REPO"
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== smoke";   timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee "$OUT/smoke.log"
echo "== pytest";  timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee "$OUT/pytest_gpu.log"
echo "== sweep";   timeout 900 python tools/bench_ops.py ${QUICK:+--quick} --json "$OUT/bench_ops.json" 2>&1 | tee "$OUT/bench_ops.log"
echo "== bench";   timeout 600 python bench.py 2>&1 | tail -3 | tee "$OUT/bench.log"
cd /tmp && export TMPDIR=/tmp
echo "== rocprof kernel trace"
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/prof_trace" -o bench -- python "$REPO/bench.py" --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/prof_trace.log" 2>&1
tail -2 "$OUT/prof_trace.log"
for CTR in FETCH_SIZE WRITE_SIZE; do
  echo "== rocprof pmc $CTR"
  timeout 600 rocprofv3 --pmc $CTR --kernel-trace -d "$OUT/prof_pmc_$CTR" -o bench -- python "$REPO/bench.py" --steps 5 --warmup 2 --no-cpu-baseline > "$OUT/prof_pmc_$CTR.log" 2>&1
  tail -2 "$OUT/prof_pmc_$CTR.log"
done
cd "$REPO"
# keep only the summaries (the raw traces can be large)
find "$OUT" -name "*.csv" -size +8M -delete
ls -R "$OUT" | head -50
