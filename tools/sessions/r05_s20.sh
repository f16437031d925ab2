#!/bin/bash
# Round 5, GPU session 20: an experiment -- the owner kernel's fill epilogue WITHOUT its in-tile fill (every hole pending: no hole
# list, no staged planes, no walk loop, no read-back; proj_fill_pending does them all): same results? what does the call gain?
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05_s20
mkdir -p "$OUT"
cd "$REPO"
LIB=memc-net_amd/lib/libmemc_hip.so
VAR=tools/probes/variants/libmemc_hip_allpend.so
timeout 200 python tools/probes/cmp_libs.py $LIB $VAR 2>&1 | grep -v amdgpu.ids | tee $OUT/allpend.txt
for ARGS in "--pan 0" "--pan 40" "--scale 2"; do
  echo "== $ARGS" | tee -a $OUT/allpend.txt
  timeout 300 python tools/ab_libs.py $LIB $VAR --op proj_fill,depth_fill --rounds 6 $ARGS 2>&1 | grep -v amdgpu.ids | tee -a $OUT/allpend.txt
done
