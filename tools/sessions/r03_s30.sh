#!/bin/bash
# Round 3, GPU session 30: HBM traffic (PMC) of the context warp's 64 x 32 / 512-lane arms (32: strips, 35: stripes of 4) next to the product.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r03_s30
mkdir -p "$OUT"
cd "$REPO"
timeout 900 python tools/pmc_any.py --out "$OUT" --match fi_fwd_tiled_c4n -- --only fi_fwd --ctx-only --variants=-1,32,35 2>&1 | grep -v amdgpu.ids | tail -8 | tee "$OUT/pmc_ctx64_512.txt"
