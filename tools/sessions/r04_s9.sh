#!/bin/bash
# Round 4, GPU session 9: kernel trace of the projection calls on the current tree (which kernel costs what).
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04_s9
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for kind in smooth iid; do
  echo "== kernel trace of the projection calls, flow=$kind"
  timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/prof_$kind" -o proj -- python "$REPO/tools/probes/proj_calls.py" $kind 60 2>&1 | grep "flow=" | tee -a "$OUT/proj_calls.txt"
  python "$REPO/tools/probes/proj_calls_summary.py" "$OUT/prof_$kind/proj_results.db" 150 | tee -a "$OUT/proj_calls.txt"
  rm -rf "$OUT/prof_$kind"
done
