#!/bin/bash
# Round 3, GPU session 6: bilinear backward at 4 workgroups per CU; the hole filler's grid; config-4 shard test.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r03_s6
mkdir -p "$OUT"
cd "$REPO"
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== config-4 shard test"
timeout 600 python -m pytest tests/test_gpu_network.py -m gpu -q -x -s -k "config4" 2>&1 | grep "pair\|passed\|failed\|Error" | tee "$OUT/pytest_config4.log"
echo "== RGB bilinear backward A/B"
timeout 600 python tools/ab_bl_bwd.py 2>&1 | grep -v amdgpu.ids | tee "$OUT/ab_bl_bwd.log"
echo "== projection: filler grid = tiles (-11) / min(tiles, 16384) (-12) / default"
timeout 900 python tools/bench_ops.py --only proj --quick --proj-variants=-11,-12 --json "$OUT/bench_proj.json" 2>&1 | grep -v "amdgpu.ids\|bwd" | tee "$OUT/bench_proj.log"
ls "$OUT"
