#!/bin/bash
# Round 3, GPU session 7: the tests added after session r03a, then the whole suite once more (final record).
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r03_s7
mkdir -p "$OUT"
cd "$REPO"
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_line.py -m gpu -q -x -k "random_shapes or plumbing or bench_line" 2>&1 | tail -15 | tee "$OUT/pytest_new.log"
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -8 | tee "$OUT/pytest_gpu.log"
cp gpurun_out/parity_errors.json "$OUT/parity_errors.json" 2>/dev/null
timeout 600 python bench.py 2>&1 | tail -1 > "$OUT/bench_line.json"
ls "$OUT"
