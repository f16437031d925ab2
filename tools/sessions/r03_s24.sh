#!/bin/bash
# Round 3, GPU session 24: the context warp with one lane per tile column on the unswizzled pixel-quad image (arm 40).
# (Arm 40 was removed after sessions 22-24: the script is the record of what ran, it no longer selects that kernel.)
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r03_s24
mkdir -p "$OUT"
cd "$REPO"
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "tile_shape_arms" 2>&1 | tail -15 | tee "$OUT/pytest.log"
timeout 600 python tools/bench_ops.py --only fi_fwd --ctx-only --ctx-flows --variants=40,-1 --json "$OUT/bench_ctx64.json" 2>&1 | grep -v amdgpu.ids | tee "$OUT/bench_ctx64.log"
sed -e 's/r03_s23/r03_s24/g' tools/sessions/r03_s23.sh | tail -n +8 > /tmp/sq.sh
bash /tmp/sq.sh
