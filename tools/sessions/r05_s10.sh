#!/bin/bash
# Round 5, GPU session 10: ragged widths on the RGB backward passes and the bilinear warp (whole quads on the tiled kernels,
# the columns behind them on the one-lane-per-site kernels): parity tests, what it costs, the aligned shapes unchanged.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05_s10
mkdir -p "$OUT"
cd "$REPO"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_workspace_and_streams.py -q -m gpu -k "ragged or unaligned or documented or multiples or interpolation or Interpolation or backward or stalled" 2>&1 | tail -6 | tee $OUT/pytest.log
timeout 400 python tools/probes/slow_paths.py 2>&1 | grep -v amdgpu.ids | tee $OUT/slow_paths.txt
