#!/bin/bash
# Round 6, GPU session 4: the projection's speculative m = 0 scan (proj_owner5 SPEC): every projection test, then in ONE process
# the working tree against round 5's and round 4's kernels (benchmark flow, pans, flow x 2), and SPEC against round 5's order
# inside the measurement build (variant -47).
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r06_s4
mkdir -p "$OUT"
cd "$REPO"
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests -q -m gpu -x -k "projection or hole or pan or far or capture or streams or workspace or concurrent" 2>&1 | tail -4 | tee "$OUT/pytest.log"
V=tools/probes/variants
for extra in "" "--pan 40" "--scale 2" "--pan 8" "--pan 160"; do
  for other in round5 round4_kernels; do
    timeout 600 python tools/ab_libs.py memc-net_amd/lib/libmemc_hip.so $V/libmemc_hip_$other.so $extra --rounds 6 2>&1 | grep -v amdgpu.ids | sed "s/^/[$extra] /" | tee -a "$OUT/proj_spec_ab_libs.txt"
  done
done
echo "== measurement build: SPEC (-1) against round 5's order (-47)"
for extra in "" "--pan 40" "--scale 2"; do
  timeout 600 python tools/ab_variants.py --op projection --variants=-1,-47 --cases proj,proj_fill,depth_fill $extra 2>&1 | grep -v amdgpu.ids | sed "s/^/[$extra] /" | tee -a "$OUT/proj_spec_ab_variants.txt"
done
timeout 300 python tools/probes/proj_small_pans.py 2>&1 | grep -v amdgpu.ids | tee "$OUT/proj_small_pans.txt"
timeout 300 python tools/probes/proj_motion_sweep.py 2>&1 | grep -v amdgpu.ids | tee "$OUT/proj_motion_sweep.txt"
timeout 600 python tools/stress_projection.py 40 2>&1 | tail -3 | tee "$OUT/stress.log"
