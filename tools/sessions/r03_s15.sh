#!/bin/bash
# Round 3, GPU session 15: small grids -- the second workgroup of every CU started 4 / 8 / 16 us late (arms 27 / 26 / 25).
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r03_s15
mkdir -p "$OUT"
cd "$REPO"
timeout 600 python tools/bench_ops.py --only fi_bwd --quick --bwd-variants 27,26,25 --json "$OUT/bench_c2_stagger.json" 2>&1 | grep -v amdgpu.ids | tee "$OUT/bench_c2_stagger.log"
