#!/bin/bash
# Round 5, GPU session 13: proj_fill_pending draws the flagged tiles from per-workgroup lists (one round of 1024 workgroups)
# instead of owning tiles g and g + 8192: projection tests, A/B against the build before it on the benchmark's flow, flow x 2
# and pans of 8 / 40 / 160 px, the motion sweep, kernel traces at pan 40 and on the benchmark's flow.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05_s13
mkdir -p "$OUT"
cd "$REPO"
LIB=memc-net_amd/lib/libmemc_hip.so
OLD=tools/probes/variants/libmemc_hip_before_fill_list.so
timeout 900 python -m pytest tests/test_gpu_workspace_and_streams.py tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py tests/test_gpu_reference.py -q -m gpu -k "workspace or graph or streams or thread or projection or Projection or pan or hole or far or ragged or multiples or config3 or stalled" 2>&1 | tail -3 | tee $OUT/pytest_proj.log
for ARGS in "--pan 0" "--scale 2" "--pan 8" "--pan 40" "--pan 160"; do
  echo "== $ARGS" | tee -a $OUT/ab_fill_list.txt
  timeout 300 python tools/ab_libs.py $OLD $LIB --op proj_fill,depth_fill --rounds 6 $ARGS 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab_fill_list.txt
done
timeout 300 python tools/probes/proj_motion_sweep.py 2>&1 | grep -v amdgpu.ids | head -17 | tee $OUT/proj_motion_sweep.txt
cd /tmp && export TMPDIR=/tmp
for ARGS in "1.0 40 1" "1.0 0 1"; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/t -o r -- python $REPO/tools/probes/proj_far_load.py $ARGS > $OUT/t.log 2>&1
  echo "scale pan fill = $ARGS" | tee -a $OUT/proj_traces.txt
  python $REPO/tools/prof_summary.py stats $OUT/t/r_results.db 2>/dev/null | head -4 | tee -a $OUT/proj_traces.txt
  rm -rf $OUT/t
done
