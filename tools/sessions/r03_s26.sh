#!/bin/bash
# Round 3, GPU session 26: final check of the tree as committed (full suite, smoke, bench line with the added secondary rows).
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r03_s26
mkdir -p "$OUT"
cd "$REPO"
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee "$OUT/smoke.log"
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -8 | tee "$OUT/pytest_gpu.log"
cp gpurun_out/parity_errors.json "$OUT/parity_errors.json" 2>/dev/null
timeout 600 python bench.py 2>&1 | tail -1 > "$OUT/bench_line.json"
ls "$OUT"
