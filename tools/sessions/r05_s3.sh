#!/bin/bash
# Round 5, GPU session 3: (a) round 4's failure with the instrumented failing variant (which scratch blocks, how long the near
# call took, what the wrong cells hold); (b) the current tree: new tests, the whole suite; (c) current kernels against round 4's
# in one process; (d) what unaligned views cost now.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05_s3
mkdir -p "$OUT"
cd "$REPO"
LIB=memc-net_amd/lib/libmemc_hip.so
cp $LIB /tmp/libmemc_hip.current.so
cp tools/probes/variants/libmemc_hip_farArm_oldScratch.so $LIB
echo "=== farArm_oldScratch, instrumented" | tee -a $OUT/variants.txt
timeout 400 python tools/probes/far_spill_streams.py --rounds 24 --product farArm_oldScratch --out $OUT/variants.txt 2>&1 | grep -v amdgpu.ids | tail -40
for i in 1 2 3; do
  timeout 300 python -m pytest tests/test_gpu_parity.py -q -k "not shifted_by_the_dominant and not dealt_out and not unaligned_views and not documented_kernel_paths" 2>&1 | tail -3 | tee -a $OUT/variants.txt
done
cp /tmp/libmemc_hip.current.so $LIB
echo "== new tests (current tree)"
timeout 900 python -m pytest tests/test_gpu_workspace_and_streams.py -q -m gpu 2>&1 | tail -15 | tee $OUT/pytest_new.log
echo "== full gpu suite"
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -25 | tee $OUT/pytest_gpu.log
cp gpurun_out/parity_errors.json $OUT/parity_errors.json 2>/dev/null || true
echo "== A/B current vs round-4 kernels"
timeout 400 python tools/ab_libs.py $LIB tools/probes/variants/libmemc_hip_round4_kernels.so --op proj,proj_fill,depth_fill --rounds 8 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_current_vs_round4.txt
echo "== slow paths"
timeout 400 python tools/probes/slow_paths.py 2>&1 | grep -v amdgpu.ids | tee $OUT/slow_paths.txt
