#!/bin/bash
# Round 3, GPU session 20: timing arms of the context warp (wrong results): 36 / 38 = no gathers (256 / 512 lanes), 37 / 39 = no
# image loads after the first chunk.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r03_s20
mkdir -p "$OUT"
cd "$REPO"
timeout 600 python tools/bench_ops.py --only fi_fwd --ctx-only --variants=-1,36,37,32,38,39 --json "$OUT/bench_ctx64.json" 2>&1 | grep -v amdgpu.ids | tee "$OUT/bench_ctx64.log"
