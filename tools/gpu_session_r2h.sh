#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r02h
mkdir -p "$OUT"
cd "$REPO"
echo "== upsample tests"; timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "upsample" 2>&1 | tail -5 | tee "$OUT/pytest_up.log"
echo "== network tests"; timeout 900 python -m pytest tests/test_gpu_network.py -m gpu -q 2>&1 | tail -5 | tee "$OUT/pytest_net.log"
echo "== bench"; timeout 600 python tools/bench_ops.py --only prologue --json "$OUT/bench_prologue.json" 2>&1 | grep -v "^$" | tee "$OUT/bench_prologue.log" | cut -c1-160
