#!/bin/bash
# Round 3, GPU session 28: the hole filler takes the counts its walks stop at from LDS when they end inside the tile (no gain in the in-process A/B of session 29; reverted).
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r03_s28
mkdir -p "$OUT"
cd "$REPO"
timeout 900 python -m pytest tests -m gpu -x -q -k "projection or Projection or fill or pan or config3 or config_3" 2>&1 | tail -4 | tee "$OUT/pytest.log"
timeout 600 python tools/bench_ops.py --only proj --json "$OUT/bench_proj.json" 2>&1 | grep -v amdgpu.ids | tee "$OUT/bench_proj.log"
timeout 300 python tools/stress_projection.py 30 2>&1 | tail -2 | tee "$OUT/stress.log"
