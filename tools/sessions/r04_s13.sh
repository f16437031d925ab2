#!/bin/bash
# Round 4, GPU session 13: the tile walk's divisions by host-made reciprocals (WalkPlan) in every tiled kernel: previous build (A)
# against this one (B) in one process; parity of everything.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04_s13
mkdir -p "$OUT"
cd "$REPO"
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -4 | tee "$OUT/pytest.log"
timeout 900 python tools/ab_libs.py memc-net_amd/lib/libmemc_hip_prev.so memc-net_amd/lib/libmemc_hip.so --op fi_fwd,fi_fwd_c2,fi_bwd,interp_fwd,interp_bwd,proj_bwd,depth_bwd,proj,proj_fill 2>&1 | grep -v amdgpu.ids | tee "$OUT/ab_walkplan.txt"
