#!/bin/bash
# Round 4, GPU session 15: the owner kernel's stripe width (tile columns per XCD stripe; the product uses 4, tuned on round 3's kernel).
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04_s15
mkdir -p "$OUT"
cd "$REPO"
timeout 900 python tools/ab_variants.py --op projection --variants=-1,110,112,115,118,104 --cases proj,proj_fill,depth --flows smooth 2>&1 | grep -v amdgpu.ids | tee "$OUT/ab_stripe_width.txt"
