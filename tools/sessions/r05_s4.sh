#!/bin/bash
# Round 5, GPU session 4: (a) what a private memory pool hands two streams that both hold their allocation (the root of round
# 4's failure?) + the instrumented failing variant with the stream handles; (b) the motion estimate on the DPP path: current
# kernels against round 4's in one process; (c) the depth operator's fixed-point timing arm; (d) the new tests.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05_s4
mkdir -p "$OUT"
cd "$REPO"
LIB=memc-net_amd/lib/libmemc_hip.so
timeout 60 tools/probes/scratch_streams 0 2>&1 | tee $OUT/pool_probe.txt
cp $LIB /tmp/libmemc_hip.current.so
cp tools/probes/variants/libmemc_hip_farArm_oldScratch.so $LIB
timeout 300 python tools/probes/far_spill_streams.py --rounds 4 --product farArm_oldScratch --out $OUT/variants.txt 2>&1 | grep -v amdgpu.ids | cut -c1-900 | head -12
cp /tmp/libmemc_hip.current.so $LIB
echo "== new tests (current tree)"
timeout 900 python -m pytest tests/test_gpu_workspace_and_streams.py tests/test_gpu_parity.py -q -m gpu -k "workspace or graph or streams or thread or projection or pan or hole or far or unaligned or documented" 2>&1 | tail -6 | tee $OUT/pytest_new.log
echo "== A/B current vs round-4 kernels"
timeout 400 python tools/ab_libs.py $LIB tools/probes/variants/libmemc_hip_round4_kernels.so --op proj,proj_fill,depth_fill --rounds 8 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_current_vs_round4.txt
echo "== depth: fixed-point timing arm"
timeout 400 python tools/ab_variants.py --op projection --variants=-1,-46 --cases depth,depth_fill --rounds 8 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_depth_fix64.txt
