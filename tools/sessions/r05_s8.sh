#!/bin/bash
# Round 5, GPU session 8: the tree as it stands -- new tests, the whole -m gpu suite, the bench line, current kernels against
# round 4's in one process, the projection's motion / pan sweep and kernel traces.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05_s8
mkdir -p "$OUT"
cd "$REPO"
LIB=memc-net_amd/lib/libmemc_hip.so
echo "== new tests"
timeout 900 python -m pytest tests/test_gpu_workspace_and_streams.py -q -m gpu 2>&1 | tail -6 | tee $OUT/pytest_new.log
echo "== full gpu suite"
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -12 | tee $OUT/pytest_gpu.log
cp gpurun_out/parity_errors.json $OUT/parity_errors.json 2>/dev/null || true
echo "== A/B current vs round-4 kernels"
timeout 400 python tools/ab_libs.py $LIB tools/probes/variants/libmemc_hip_round4_kernels.so --op proj,proj_fill,depth_fill --rounds 8 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_current_vs_round4.txt
echo "== bench"
timeout 600 python bench.py > $OUT/bench.log 2>$OUT/bench.err; tail -c 2500 $OUT/bench.log
echo "== projection rows"
timeout 300 python tools/probes/proj_motion_sweep.py 2>&1 | grep -v amdgpu.ids | tail -25 | tee $OUT/proj_motion_sweep.txt
echo "== traces"
cd /tmp && export TMPDIR=/tmp
for ARGS in "2.0 0 1" "1.0 40 1" "1.0 0 1"; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/t -o r -- python $REPO/tools/probes/proj_far_load.py $ARGS > $OUT/t.log 2>&1
  echo "scale pan fill = $ARGS" | tee -a $OUT/proj_traces.txt
  python $REPO/tools/prof_summary.py stats $OUT/t/r_results.db 2>/dev/null | head -5 | tee -a $OUT/proj_traces.txt
  rm -rf $OUT/t
done
