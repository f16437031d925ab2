#!/bin/bash
# Round 5, GPU session 7: round 4's failing configuration rebuilt CONSISTENTLY (sessions 2-6 mixed two versions of the scratch
# header in one library: fi_bwd_cn.hip includes it too -- what those sessions ran was one block shared by all streams without
# ordering): every translation unit on round 4's header, the spilling far kernel as the product's.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05_s7
mkdir -p "$OUT"
cd "$REPO"
LIB=memc-net_amd/lib/libmemc_hip.so
cp $LIB /tmp/libmemc_hip.current.so
for V in farArm_oldScratch product_oldScratch; do
  cp tools/probes/variants/libmemc_hip_$V.so $LIB
  echo "=== $V (round 4's cache in every translation unit)" | tee -a $OUT/variants.txt
  timeout 300 python tools/probes/far_spill_streams.py --rounds 16 --product $V --out $OUT/variants.txt 2>&1 | grep -v amdgpu.ids | cut -c1-1000 | head -16
  for i in 1 2 3; do
    timeout 300 python -m pytest tests/test_gpu_parity.py -q -k "not shifted_by_the_dominant and not dealt_out and not unaligned_views and not documented_kernel_paths and not multiples_of_four" 2>&1 | tail -2 | tee -a $OUT/variants.txt
  done
done
cp /tmp/libmemc_hip.current.so $LIB
