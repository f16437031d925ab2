#!/bin/bash
# Round 3, GPU session 27: FlowProjection with hole filling -- tiles whose every cell has a positive count write the trivial
# summary after one vote (no LDS atomics, no row reductions): parity, then time against the previous build's numbers.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r03_s27
mkdir -p "$OUT"
cd "$REPO"
timeout 900 python -m pytest tests -m gpu -x -q -k "projection or Projection or fill or pan or config3 or config_3" 2>&1 | tail -4 | tee "$OUT/pytest.log"
timeout 600 python tools/bench_ops.py --only proj --json "$OUT/bench_proj.json" 2>&1 | grep -v amdgpu.ids | tee "$OUT/bench_proj.log"
timeout 300 python tools/stress_projection.py 30 2>&1 | tail -2 | tee "$OUT/stress.log"
