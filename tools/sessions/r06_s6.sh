#!/bin/bash
# Round 6, GPU session 6: the motion estimate through the scalar unit (measurement arm -50) against the product (-1), no estimate
# (-48) and 16 per-lane samples (-49), one process.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r06_s6
mkdir -p "$OUT"
cd "$REPO"
for r in 1 2; do
timeout 600 python tools/ab_variants.py --op projection --variants=-1,-48,-49,-50 --cases proj,proj_fill,depth,depth_fill --rounds 8 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/proj_motion_estimate_scalar_arm.txt"
done
