#!/bin/bash
# Round 6, GPU session 17: the motion estimate cached per image in the call's scratch (measurement arm -54) against the product
# (-1) and no estimate at all (-48), one process, two runs.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r06_s17
mkdir -p "$OUT"
cd "$REPO"
for r in 1 2; do
timeout 600 python tools/ab_variants.py --op projection --variants=-1,-48,-54 --cases proj,proj_fill,depth,depth_fill --rounds 8 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/proj_motion_cache_arm.txt"
done
timeout 600 python tools/ab_variants.py --op projection --variants=-1,-54 --cases proj,proj_fill,depth_fill --rounds 6 --pan 40 2>&1 | grep -v amdgpu.ids | sed "s/^/[--pan 40] /" | tee -a "$OUT/proj_motion_cache_arm.txt"
