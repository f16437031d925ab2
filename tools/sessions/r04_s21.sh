#!/bin/bash
# Round 4, GPU session 21: where does a panned projection spend its time?  Kernel trace of FlowProjection with hole
# filling on the benchmark's flow + a pan of 40 px, and without filling.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04_s21
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for FILL in 1 0; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/t$FILL -o r -- python $REPO/tools/probes/proj_far_load.py 1 40 $FILL > $OUT/t$FILL.log 2>&1
  python $REPO/tools/prof_summary.py stats $OUT/t$FILL/r_results.db 2>/dev/null | head -12 | tee $OUT/pan40_fill$FILL.txt
  ls $OUT/t$FILL | head
done
