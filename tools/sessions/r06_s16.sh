#!/bin/bash
# Round 6, GPU session 16: the RGB FilterInterpolation backward at ragged widths with the aligned kernel body (sites that touch the
# partial quad take the per-site path): tests, then the slow-paths probe (aligned / width 1278 / unaligned view, every operator).
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r06_s16
mkdir -p "$OUT"
cd "$REPO"
timeout 1500 python -m pytest tests -q -m gpu -x -k "ragged or rgb or random_shapes or random_strided or documented or unaligned or multiples" 2>&1 | tail -4 | tee "$OUT/pytest.log"
timeout 600 python tools/probes/slow_paths.py 2>&1 | grep -v amdgpu.ids | tee "$OUT/slow_paths.txt"
