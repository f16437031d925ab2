#!/bin/bash
# Round 5, GPU session 24: the projection forward behind a caller's zero fill of its outputs (the reference's contract; the shipped
# Python layer does not zero-fill): tools/bench_ops.py shows single calls at 159 / 182 / 240 us there against 115 / 154 / 226 without.
# Is it the streaming stores meeting the memset's dirty lines?  Product (streaming stores) against a variant with cached stores,
# with and without the zero fill.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05_s24
mkdir -p "$OUT"
cd "$REPO"
LIB=memc-net_amd/lib/libmemc_hip.so
VAR=tools/probes/variants/libmemc_hip_proj_cached_stores.so
for ARGS in "" "--prezero"; do
  echo "== A = streaming stores (product), B = cached stores; $ARGS" | tee -a $OUT/prezero.txt
  timeout 300 python tools/ab_libs.py $LIB $VAR --op proj,proj_fill,depth_fill --rounds 6 $ARGS 2>&1 | grep -v amdgpu.ids | tee -a $OUT/prezero.txt
done
