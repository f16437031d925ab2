#!/bin/bash
# Round 5, GPU session 6: the failing variant, which stream handle the LIBRARY saw and which cache entry it took.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05_s6
mkdir -p "$OUT"
cd "$REPO"
LIB=memc-net_amd/lib/libmemc_hip.so
cp $LIB /tmp/libmemc_hip.current.so
cp tools/probes/variants/libmemc_hip_farArm_oldScratch.so $LIB
timeout 300 python tools/probes/far_spill_streams.py --rounds 2 --product farArm_oldScratch --out $OUT/variants.txt 2>&1 | grep -v amdgpu.ids | cut -c1-900 | head -5
cp /tmp/libmemc_hip.current.so $LIB
