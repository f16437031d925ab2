#!/bin/bash
# Round 5, GPU session 25: the owner kernel's phases per workgroup at the benchmark's flow and at twice / three times the flow.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05_s25
mkdir -p "$OUT"
cd "$REPO"
timeout 300 python tools/probes/proj_pan_phases.py 1:0 1.5:0 2:0 2>&1 | grep -v amdgpu.ids | tee $OUT/proj_scale_phases.txt
