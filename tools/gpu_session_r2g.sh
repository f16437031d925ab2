#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r02g
mkdir -p "$OUT"
cd "$REPO"
echo "== ctx tests"; timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "ctx or blend" 2>&1 | tail -12 | tee "$OUT/pytest_ctx.log"
echo "== network tests"; timeout 900 python -m pytest tests/test_gpu_network.py -m gpu -q 2>&1 | tail -5 | tee "$OUT/pytest_net.log"
echo "== bench ctx"; timeout 600 python tools/bench_ops.py --only fi_ctx,fi_blend --json "$OUT/bench_ctx.json" 2>&1 | grep -v "^$" | tee "$OUT/bench_ctx.log" | cut -c1-160
