#!/bin/bash
# Round 3, GPU session 25: RGB bilinear warp FORWARD on 64 x 32 tiles / 512 lanes (bl_cap 4 / 5); the backward after the refactoring.
# (The forward arms bl_cap 4 / 5 were removed after this session; bl_cap 4 now only selects the 64 x 16 / 256-lane backward.)
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r03_s25
mkdir -p "$OUT"
cd "$REPO"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference.py -x -q -k "interpolation or Interpolation or tile_walks or bilinear or warp" 2>&1 | tail -5 | tee "$OUT/pytest.log"
timeout 600 python tools/ab_bl_bwd.py 2>&1 | grep -v amdgpu.ids | tee "$OUT/ab_bl.log"
