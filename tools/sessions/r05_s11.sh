#!/bin/bash
# Round 5, GPU session 11: the image's motion posted BEHIND the first barrier and polled: A/B against round 4's kernels,
# projection tests, pan rows.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05_s11
mkdir -p "$OUT"
cd "$REPO"
LIB=memc-net_amd/lib/libmemc_hip.so
timeout 900 python -m pytest tests/test_gpu_workspace_and_streams.py tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py -q -m gpu -k "workspace or graph or streams or thread or projection or Projection or pan or hole or far or ragged or multiples or config3 or stalled" 2>&1 | tail -3 | tee $OUT/pytest_proj.log
timeout 400 python tools/ab_libs.py $LIB tools/probes/variants/libmemc_hip_round4_kernels.so --op proj,proj_fill,depth_fill --rounds 8 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_current_vs_round4.txt
timeout 300 python tools/probes/proj_motion_sweep.py 2>&1 | grep -v amdgpu.ids | head -17 | tee $OUT/proj_motion_sweep.txt
