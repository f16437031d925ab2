#!/bin/bash
# tools/probes/variants/build_variants.sh -- rebuilds the variant PRODUCT libraries round 5 used to take round 4's failure apart
# (sessions tools/sessions/r05_s2.sh ... r05_s7.sh) from the repository's own history:
#   * the commit of round 5's first GPU session (round 4's kernels + the workspace entry points + the spilling far kernel as a
#     template instantiation), checked out into a scratch worktree;
#   * variants_on_session1_commit.patch: -DMEMC_FAR_ARM (the spilling arm as the product's far kernel), -DMEMC_OLD_SCRATCH
#     (round 4's memc_scratch.hpp, taken from round 4's last commit, in EVERY translation unit that includes it -- sessions 2-6
#     switched it in flow_projection.hip only: fi_bwd_cn.hip kept the new header, one CallScratch::alloc survived the link and
#     every stream got ONE unordered block, which is what reproduced round 4's symptom), -DMEMC_DEBUG_BLOCK (which block /
#     stream / cache entry the thread's last projection call used).
# Output: libmemc_hip_{farArm_oldScratch,farArm_newScratch,product_oldScratch,round4_kernels}.so next to this script
# (git-ignored binaries; they travel to the GPU box with the snapshot).
set -eu
HERE=$(cd "$(dirname "$0")" && pwd)
REPO=$(cd "$HERE/../../.." && pwd)
SESSION1_COMMIT=${SESSION1_COMMIT:-f0d3781}
ROUND4_COMMIT=${ROUND4_COMMIT:-4c931c1}
WT=$(mktemp -d)
git -C "$REPO" worktree add -f "$WT" "$SESSION1_COMMIT" >/dev/null
trap 'git -C "$REPO" worktree remove --force "$WT"' EXIT
cd "$WT"
git apply "$HERE/variants_on_session1_commit.patch"
git -C "$REPO" show "$ROUND4_COMMIT:memc-net_amd/csrc/memc_scratch.hpp" > memc-net_amd/csrc/memc_scratch_r4.hpp
cd memc-net_amd/csrc
SR="filter_interpolation.hip fi_bwd_c3.hip fi_bwd_cn.hip interpolation.hip flow_projection.hip flow_prologue.hip calibration.hip layer_api.cpp"
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -munsafe-fp-atomics -I../../include -shared"
/opt/rocm/bin/hipcc $FL -DMEMC_FAR_ARM -DMEMC_OLD_SCRATCH -DMEMC_DEBUG_BLOCK -o "$HERE/libmemc_hip_farArm_oldScratch.so" $SR
/opt/rocm/bin/hipcc $FL -DMEMC_FAR_ARM -o "$HERE/libmemc_hip_farArm_newScratch.so" $SR
/opt/rocm/bin/hipcc $FL -DMEMC_OLD_SCRATCH -o "$HERE/libmemc_hip_product_oldScratch.so" $SR
/opt/rocm/bin/hipcc $FL -o "$HERE/libmemc_hip_round4_kernels.so" $SR
ls -la "$HERE"/*.so
