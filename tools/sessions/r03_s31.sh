#!/bin/bash
# Round 3, GPU session 31: the context warp on 128 x 8 tiles (arms 40 / 41): parity, time, HBM traffic (PMC).
# (The arms were removed after this session: the script is the record of what ran.)
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r03_s31
mkdir -p "$OUT"
cd "$REPO"
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "tile_shape_arms" 2>&1 | tail -3 | tee "$OUT/pytest.log"
timeout 600 python tools/bench_ops.py --only fi_fwd --ctx-only --variants=-1,40,41,31,-1,40 --json "$OUT/bench_ctx64.json" 2>&1 | grep -v amdgpu.ids | tee "$OUT/bench_ctx64.log"
timeout 900 python tools/pmc_any.py --out "$OUT" --match fi_fwd_tiled_c4n -- --only fi_fwd --ctx-only --variants=-1,40,41 2>&1 | grep -v amdgpu.ids | tail -6 | tee "$OUT/pmc_ctx64_lx32.txt"
