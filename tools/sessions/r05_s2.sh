#!/bin/bash
# Round 5, GPU session 2: (a) round 4's failing configuration rebuilt from the code of session 1 -- the spilling far kernel
# as the PRODUCT's far kernel, with round 4's per-stream-handle scratch cache and with round 5's event-ordered blocks -- under
# the very test that failed, in its own pytest processes, in the whole test file, and dissected; (b) the shifted scan, the
# dealt-out far tiles, the device-side tag of captured calls: new tests, projection tests, bench rows, traces.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05_s2
mkdir -p "$OUT"
cd "$REPO"
LIB=memc-net_amd/lib/libmemc_hip.so
cp $LIB /tmp/libmemc_hip.current.so
for V in farArm_oldScratch farArm_newScratch product_oldScratch; do
  cp tools/probes/variants/libmemc_hip_$V.so $LIB
  echo "=== variant $V" | tee -a $OUT/variants.txt
  for i in 1 2 3 4; do
    timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -k "concurrent_streams" 2>&1 | tail -1 | tee -a $OUT/variants.txt
  done
  timeout 300 python tools/probes/far_spill_streams.py --rounds 10 --product $V --out $OUT/variants.txt 2>&1 | grep -v amdgpu.ids | tail -30
done
cp tools/probes/variants/libmemc_hip_farArm_oldScratch.so $LIB
echo "=== the whole test_gpu_parity.py on farArm_oldScratch (old tests only)" | tee -a $OUT/variants.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -q -k "not shifted_by_the_dominant and not dealt_out" 2>&1 | tail -8 | tee -a $OUT/variants.txt
cp /tmp/libmemc_hip.current.so $LIB
echo "== new tests (current tree)"
timeout 900 python -m pytest tests/test_gpu_workspace_and_streams.py tests/test_gpu_parity.py -q -m gpu -k "workspace or graph or streams or thread or projection or pan or hole or far" 2>&1 | tail -25 | tee $OUT/pytest_proj.log
echo "== projection rows"
timeout 300 python tools/probes/proj_motion_sweep.py 2>&1 | tail -25 | tee $OUT/proj_motion_sweep.txt
echo "== traces"
cd /tmp && export TMPDIR=/tmp
for ARGS in "2.0 0 1" "1.0 40 1" "1.0 0 1" "1.0 0 0"; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/t -o r -- python $REPO/tools/probes/proj_far_load.py $ARGS > $OUT/t.log 2>&1
  echo "scale pan fill = $ARGS" | tee -a $OUT/proj_traces.txt
  python $REPO/tools/prof_summary.py stats $OUT/t/r_results.db 2>/dev/null | head -5 | tee -a $OUT/proj_traces.txt
  rm -rf $OUT/t
done
