#!/bin/bash
# Round 5, GPU session 19: a dead zone of 6 px for the dominant motion (no shifted scan for a drift of a few pixels): projection
# tests, A/B against the build before at pans of 0 / 3 / 5 / 8 / 12 / 40 px and at twice the flow, the pan sweep.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05_s19
mkdir -p "$OUT"
cd "$REPO"
LIB=memc-net_amd/lib/libmemc_hip.so
OLD=tools/probes/variants/libmemc_hip_before_dead_zone.so
timeout 900 python -m pytest tests/test_gpu_workspace_and_streams.py tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py tests/test_gpu_reference.py -q -m gpu -k "workspace or graph or streams or thread or projection or Projection or pan or hole or far or ragged or multiples or config3 or stalled" 2>&1 | tail -3 | tee $OUT/pytest_proj.log
for ARGS in "--pan 0" "--pan 3" "--pan 5" "--pan 8" "--pan 12" "--pan 40" "--scale 2"; do
  echo "== $ARGS" | tee -a $OUT/ab.txt
  timeout 300 python tools/ab_libs.py $OLD $LIB --op proj,proj_fill,depth_fill --rounds 5 $ARGS 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab.txt
done
timeout 300 python tools/probes/proj_motion_sweep.py 2>&1 | grep -v amdgpu.ids | head -17 | tee $OUT/proj_motion_sweep.txt
