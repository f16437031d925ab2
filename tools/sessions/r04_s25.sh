#!/bin/bash
# Round 4, GPU session 25: kernel trace of FlowProjection + fill under an 18 px pan (flow x 0.25): owner kernel vs pending filler.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04_s25
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for ARGS in "0.25 18 1" "0.25 18 0" "0.25 0 1"; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/t -o r -- python $REPO/tools/probes/proj_far_load.py $ARGS > $OUT/t.log 2>&1
  echo "scale pan fill = $ARGS" | tee -a $OUT/pan18.txt
  python $REPO/tools/prof_summary.py stats $OUT/t/r_results.db 2>/dev/null | head -4 | tee -a $OUT/pan18.txt
  rm -rf $OUT/t
done
