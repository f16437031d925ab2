#!/bin/bash
# Round 3, GPU session 12: C=64 forward with transposed LDS rows (arms 37 / 38).
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r03_s12
mkdir -p "$OUT"
cd "$REPO"
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "tile_shape_arms" 2>&1 | tail -6 | tee "$OUT/pytest_new.log"
timeout 600 python tools/bench_ops.py --only fi_fwd --ctx-only --variants=-1,37,38,31 --json "$OUT/bench_ctx64_arms.json" 2>&1 | grep -v amdgpu.ids | tee "$OUT/bench_ctx64_arms.log"
timeout 600 bash tools/pmc_sq.sh r03_s12/sq_tr fi_fwd fi_fwd_tiled_c4n "--ctx-only --variants=37" 2>&1 | grep -v amdgpu.ids | tail -8 | tee "$OUT/sq_tr.log"
ls "$OUT"
