#!/bin/bash
# Round 6, GPU session 5: what does the motion estimate cost the benchmark's flow?  Timing arms of the measurement build in one
# process: -1 the product (round 5's order), -47 the speculative pass, -48 no estimate at all, -49 sixteen samples; then the
# product against round 4's kernels once more, and the projection tests on the product (replay counter as a device global).
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r06_s5
mkdir -p "$OUT"
cd "$REPO"
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests -q -m gpu -x -k "projection or hole or pan or far or capture or streams or workspace or concurrent or filter_interpolation_backward" 2>&1 | tail -3 | tee "$OUT/pytest.log"
for r in 1 2; do
timeout 600 python tools/ab_variants.py --op projection --variants=-1,-48,-49,-47 --cases proj,proj_fill,depth,depth_fill --rounds 8 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/proj_motion_estimate_arms.txt"
done
V=tools/probes/variants
timeout 600 python tools/ab_libs.py memc-net_amd/lib/libmemc_hip.so $V/libmemc_hip_round4_kernels.so --rounds 8 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/proj_ab_round4.txt"
timeout 600 python tools/ab_libs.py memc-net_amd/lib/libmemc_hip.so $V/libmemc_hip_round5.so --rounds 8 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/proj_ab_round5.txt"
