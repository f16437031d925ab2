#!/bin/bash
# Round 4, GPU session 19: shader-core counters of proj_owner_far at 4x the benchmark's motion (97 % of the tiles recomputed).
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"
PMC_CMD="python $REPO/tools/probes/proj_far_load.py 4" bash tools/pmc_sq.sh r04_s19/sq x "proj_owner_far" 2>&1 | tail -45 | tee gpurun_out/r04_s19_far_sq.txt
