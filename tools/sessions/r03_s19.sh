#!/bin/bash
# Round 3, GPU session 19: the context warp with two LDS buffers and one barrier per chunk (arms 36-38).
# (The two-buffer arms 36-38 were removed after this session; the numbers 36-39 now name the timing arms of session 20.)
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r03_s19
mkdir -p "$OUT"
cd "$REPO"
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "tile_shape_arms" 2>&1 | tail -3 | tee "$OUT/pytest.log"
timeout 600 python tools/bench_ops.py --only fi_fwd --ctx-only --variants=-1,36,37,38,32,-1,36,37 --json "$OUT/bench_ctx64.json" 2>&1 | grep -v amdgpu.ids | tee "$OUT/bench_ctx64.log"
