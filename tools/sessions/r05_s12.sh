#!/bin/bash
# Round 5, GPU session 12: the scan's iterations reordered (near rows first, the far rows' fx requested by an early row test):
# projection tests, A/B against round 4's kernels, traces at flow x 2 and on the benchmark's flow, the motion sweep.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05_s12
mkdir -p "$OUT"
cd "$REPO"
LIB=memc-net_amd/lib/libmemc_hip.so
timeout 900 python -m pytest tests/test_gpu_workspace_and_streams.py tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py tests/test_gpu_reference.py -q -m gpu -k "workspace or graph or streams or thread or projection or Projection or pan or hole or far or ragged or multiples or config3 or stalled" 2>&1 | tail -3 | tee $OUT/pytest_proj.log
timeout 400 python tools/ab_libs.py $LIB tools/probes/variants/libmemc_hip_round4_kernels.so --op proj,proj_fill,depth_fill --rounds 8 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_current_vs_round4.txt
timeout 300 python tools/probes/proj_motion_sweep.py 2>&1 | grep -v amdgpu.ids | head -17 | tee $OUT/proj_motion_sweep.txt
cd /tmp && export TMPDIR=/tmp
for ARGS in "2.0 0 1" "1.0 0 1"; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/t -o r -- python $REPO/tools/probes/proj_far_load.py $ARGS > $OUT/t.log 2>&1
  echo "scale pan fill = $ARGS" | tee -a $OUT/proj_traces.txt
  python $REPO/tools/prof_summary.py stats $OUT/t/r_results.db 2>/dev/null | head -4 | tee -a $OUT/proj_traces.txt
  rm -rf $OUT/t
done
