#!/bin/bash
# Round 6, GPU session 21: the RGB forward on 128 x 8 tiles (arms 20-25 of the measurement build) -- results against the product,
# then timing in one process.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r06_s21
mkdir -p "$OUT"
cd "$REPO"
timeout 300 python tools/probes/chk_fi_fwd_variants.py 20,21,22,23,24,25 2>&1 | grep -v amdgpu.ids | tee "$OUT/chk.txt" | grep -c "max|diff| 0 "
grep DIFFERS "$OUT/chk.txt" | head
for r in 1 2; do
timeout 600 python tools/ab_variants.py --op fi_fwd --variants=-1,20,21,22,23,24,25,15 --cases fi_fwd --flows smooth,iid --rounds 6 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/fi_fwd_wide_tiles_ab.txt"
done
