#!/bin/bash
# Round 4, GPU session 6: full GPU test suite on the tree; the RGB backward on 64 x 8 tiles for small grids (A/B in one process);
# what the scalar / direct kernels cost; the VE call trace.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04_s6
mkdir -p "$OUT"
cd "$REPO"
echo "== pytest -m gpu (everything)"
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -25 | tee "$OUT/pytest_gpu.log"
cp gpurun_out/parity_errors.json "$OUT/parity_errors.json" 2>/dev/null
echo "== RGB backward: 64 x 16 (60) against 64 x 8 (61) tiles, config 2 and 720p, one process"
timeout 600 python tools/ab_variants.py --op fi_bwd --variants=60,61 --cases fi_bwd_c2,fi_bwd --flows smooth,iid 2>&1 | grep -v amdgpu.ids | tee "$OUT/ab_fi_bwd_tiles.txt"
echo "== slow paths"
timeout 600 python tools/probes/slow_paths.py 2>&1 | grep -v amdgpu.ids | tee "$OUT/slow_paths.txt"
