#!/bin/bash
# Round 5, GPU session 14: where do proj_fill_pending's 40 us under a pan of 40 px go?  Timing arms of the kernel (variant
# builds, -DMEMC_FILL_ARM=n: 1 no hole loop, 2 no walks beyond the tile, 3 eight holes per lane in flight, 4 two, 5 no stores),
# each against the product in one process, whole call at pan 40 (the other kernels of the call are the same code).
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05_s14
mkdir -p "$OUT"
cd "$REPO"
LIB=memc-net_amd/lib/libmemc_hip.so
for a in 1 2 3 4 5; do
  timeout 200 python tools/ab_libs.py $LIB tools/probes/variants/libmemc_hip_fillarm$a.so --op proj_fill --rounds 5 --pan 40 2>&1 | grep -v amdgpu.ids | tee -a $OUT/fill_arms.txt
done
timeout 200 python tools/ab_libs.py $LIB tools/probes/variants/libmemc_hip_fillarm3.so --op proj_fill --rounds 5 --pan 0 2>&1 | grep -v amdgpu.ids | tee -a $OUT/fill_arms.txt
timeout 200 python tools/ab_libs.py $LIB tools/probes/variants/libmemc_hip_fillarm4.so --op proj_fill --rounds 5 --pan 0 2>&1 | grep -v amdgpu.ids | tee -a $OUT/fill_arms.txt
