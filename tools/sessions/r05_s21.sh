#!/bin/bash
# Round 5, GPU session 21: where do the owner kernel's ~20 us of hole-filling epilogue go?  Timing arms (variant builds, results
# WRONG -- holes stay unfilled): 1 no epilogue at all, 2 the vote "any hole in the tile?" (one barrier) and nothing else, 3 the
# no-hole path for every tile (vote + the trivial summaries); against the product, whole call with fill 1, benchmark flow.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05_s21
mkdir -p "$OUT"
cd "$REPO"
LIB=memc-net_amd/lib/libmemc_hip.so
for a in 1 2 3; do
  timeout 300 python tools/ab_libs.py $LIB tools/probes/variants/libmemc_hip_epi$a.so --op proj_fill,depth_fill --rounds 6 2>&1 | grep -v amdgpu.ids | tee -a $OUT/epilogue_arms.txt
done
timeout 300 python tools/ab_libs.py $LIB $LIB --op proj,proj_fill --rounds 6 2>&1 | grep -v amdgpu.ids | tee -a $OUT/epilogue_arms.txt
