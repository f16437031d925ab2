#!/bin/bash
# Round 5, GPU session 15: the owner kernel's phases per workgroup under a pan of 40 px, with and without hole filling.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05_s15
mkdir -p "$OUT"
cd "$REPO"
timeout 300 python tools/probes/proj_pan_phases.py 2>&1 | grep -v amdgpu.ids | tee $OUT/proj_pan_phases.txt
