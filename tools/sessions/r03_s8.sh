#!/bin/bash
# Round 3, GPU session 8: C=64 forward on 32x32 tiles with the row-parity swizzle.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r03_s8
mkdir -p "$OUT"
cd "$REPO"
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "tile_shape_arms or test_filter_interpolation" 2>&1 | tail -6 | tee "$OUT/pytest_new.log"
echo "== C=64 forward: 64x16 strips (-1) / 32x32 strips (33) / stripes of 4 (34) / 2 (35) / 8 (36) / 64x16 stripes of 4 (31)"
timeout 600 python tools/bench_ops.py --only fi_fwd --ctx-only --variants=-1,33,34,35,36,31 --json "$OUT/bench_ctx64_arms.json" 2>&1 | grep -v amdgpu.ids | tee "$OUT/bench_ctx64_arms.log"
echo "== SQ counters: product, then 32x32 stripes of 4"
timeout 600 bash tools/pmc_sq.sh r03_s8/sq_product fi_fwd fi_fwd_tiled_c4n "--ctx-only" 2>&1 | grep -v amdgpu.ids | tail -12 | tee "$OUT/sq_product.log"
timeout 600 bash tools/pmc_sq.sh r03_s8/sq_lx8 fi_fwd fi_fwd_tiled_c4n "--ctx-only --variants=34" 2>&1 | grep -v amdgpu.ids | tail -12 | tee "$OUT/sq_lx8.log"
ls "$OUT"
