#!/bin/bash
# Round 3, GPU session 18: the context warp on 64 x 32 tiles / 512 lanes again, this time with a staging budget that holds
# the tile's whole box (6144 cells, 96 KiB; round 2's arm had 3584 and swept most tiles in two bands).
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r03_s18
mkdir -p "$OUT"
cd "$REPO"
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "tile_shape_arms" 2>&1 | tail -3 | tee "$OUT/pytest.log"
timeout 600 python tools/bench_ops.py --only fi_fwd --ctx-only --variants=-1,32,35,31,-1,32 --json "$OUT/bench_ctx64.json" 2>&1 | grep -v amdgpu.ids | tee "$OUT/bench_ctx64.log"
