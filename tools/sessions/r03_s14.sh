#!/bin/bash
# Round 3, GPU session 14: C=64 forward with cell addresses packed once per band (arms 40 / 41).
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r03_s14
mkdir -p "$OUT"
cd "$REPO"
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "tile_shape_arms" 2>&1 | tail -4 | tee "$OUT/pytest_new.log"
timeout 600 python tools/bench_ops.py --only fi_fwd --ctx-only --variants=-1,40,41,37 --json "$OUT/bench_ctx64_arms.json" 2>&1 | grep -v amdgpu.ids | tee "$OUT/bench_ctx64_arms.log"
