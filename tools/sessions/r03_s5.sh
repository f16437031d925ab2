#!/bin/bash
# Round 3, GPU session 5: in-owner hole filling + filler grid; 32x16 tiles of the RGB backward on small grids.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r03_s5
mkdir -p "$OUT"
cd "$REPO"
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== tests that cover the changes first"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference.py -m gpu -q -x -k "projection or rgb or fillhole or pan or reference" 2>&1 | tail -8 | tee "$OUT/pytest_new.log"
echo "== the config-4 shard test that failed in session 4 (detail)"
timeout 600 python -m pytest tests/test_gpu_network.py -m gpu -q -x -k "config4" 2>&1 | grep -v "^  \|Warning" | tail -40 | tee "$OUT/pytest_config4.log"
echo "== RGB backward: tile shape by grid (-1) / 64x16 forced (24) / 32x16 forced (25)"
timeout 900 python tools/bench_ops.py --only fi_bwd --bwd-variants 24,25 --json "$OUT/bench_fi_bwd_tiles.json" 2>&1 | grep -v amdgpu.ids | tee "$OUT/bench_fi_bwd_tiles.log"
echo "== projection: filler grid = tiles (-11) / min(tiles, 16384) (-12) / default"
timeout 900 python tools/bench_ops.py --only proj --proj-variants=-11,-12 --json "$OUT/bench_proj.json" 2>&1 | grep -v amdgpu.ids | tee "$OUT/bench_proj.log"
echo "== full GPU suite"
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -8 | tee "$OUT/pytest_gpu.log"
cp gpurun_out/parity_errors.json "$OUT/" 2>/dev/null
ls "$OUT"
