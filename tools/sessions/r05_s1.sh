#!/bin/bash
# Round 5, GPU session 1: (a) the minimal private-scratch reproducer and the dissection of round 4's failing far-kernel arm
# (VERDICT r4 item 1), (b) the new workspace / HIP-graph / multi-stream tests and the whole -m gpu suite on the event-ordered
# scratch blocks, (c) bench.py with config 2 on rotating input sets, (d) kernel traces of the projection at flow x 2 and
# under a 40 px pan (baseline for the shifted scan).
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05_s1
mkdir -p "$OUT"
cd "$REPO"
echo "== scratch_streams" | tee $OUT/scratch_streams.txt
timeout 420 tools/probes/scratch_streams 10000 2>&1 | tee -a $OUT/scratch_streams.txt
echo "== far_spill_streams"
timeout 600 python tools/probes/far_spill_streams.py --rounds 8 --out $OUT/far_spill_streams.txt 2>&1 | tail -60
echo "== new tests"
timeout 900 python -m pytest tests/test_gpu_workspace_and_streams.py -x -q -m gpu 2>&1 | tail -25 | tee $OUT/pytest_new.log
echo "== full gpu suite"
timeout 1200 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 | tee $OUT/pytest_gpu.log
cp gpurun_out/parity_errors.json $OUT/parity_errors.json 2>/dev/null || true
echo "== bench"
timeout 600 python bench.py > $OUT/bench.log 2>$OUT/bench.err; tail -c 6000 $OUT/bench.log
echo "== traces"
cd /tmp && export TMPDIR=/tmp
for ARGS in "2.0 0 1" "1.0 40 1" "1.0 0 1"; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/t -o r -- python $REPO/tools/probes/proj_far_load.py $ARGS > $OUT/t.log 2>&1
  echo "scale pan fill = $ARGS" | tee -a $OUT/proj_traces.txt
  python $REPO/tools/prof_summary.py stats $OUT/t/r_results.db 2>/dev/null | head -5 | tee -a $OUT/proj_traces.txt
  rm -rf $OUT/t
done
