#!/bin/bash
# Round 4, GPU session 17: FlowProjection forward against the size of the motion (what proj_owner_far costs).
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04_s17
mkdir -p "$OUT"
cd "$REPO"
timeout 600 python tools/probes/proj_motion_sweep.py 2>&1 | grep -v amdgpu.ids | tee "$OUT/motion_sweep.txt"
