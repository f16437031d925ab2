#!/bin/bash
# Round 3, GPU session 16: RGB bilinear backward on 64 x 32 tiles / 512 lanes (bl_cap 4): fewer flushed cells per site.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r03_s16
mkdir -p "$OUT"
cd "$REPO"
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "tile_walks or packed_planes" 2>&1 | tail -5 | tee "$OUT/pytest.log"
timeout 600 python tools/ab_bl_bwd.py 2>&1 | grep -v amdgpu.ids | tee "$OUT/ab_bl_bwd.log"
