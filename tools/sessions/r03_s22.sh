#!/bin/bash
# Round 3, GPU session 22: the context warp on a planar LDS image, one lane per tile column (arm 40): parity, then time.
# (Arm 40 was removed after sessions 22-24: the script is the record of what ran, it no longer selects that kernel.)
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r03_s22
mkdir -p "$OUT"
cd "$REPO"
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "tile_shape_arms" 2>&1 | tail -15 | tee "$OUT/pytest.log"
timeout 600 python tools/bench_ops.py --only fi_fwd --ctx-only --ctx-flows --variants=40,-1 --json "$OUT/bench_ctx64.json" 2>&1 | grep -v amdgpu.ids | tee "$OUT/bench_ctx64.log"
