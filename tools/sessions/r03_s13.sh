#!/bin/bash
# Round 3, GPU session 13: is the transposed-layout arm slower because of its LDS request (occupancy) or its code?
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r03_s13
mkdir -p "$OUT"
cd "$REPO"
timeout 600 python tools/bench_ops.py --only fi_fwd --ctx-only --variants=-1,39,37 --json "$OUT/bench_ctx64_arms.json" 2>&1 | grep -v amdgpu.ids | tee "$OUT/bench_ctx64_arms.log"
