#!/bin/bash
# Round 4, GPU session 11: the bench line on the tree (traffic from the committed PMC record of these sources, secondary rows warmed by time).
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04_s11
mkdir -p "$OUT"
cd "$REPO"
timeout 900 python bench.py 2>&1 | tail -1 | tee "$OUT/bench_line.json"
timeout 600 python -m pytest tests/test_gpu_bench_line.py -m gpu -q 2>&1 | tail -3
