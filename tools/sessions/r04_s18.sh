#!/bin/bash
# Round 4, GPU session 18: proj_owner_far rewritten (reach from the recorded bounds, only the tiles a far source can reach,
# direct splats, two workgroups per CU) -- parity, stress, the motion sweep again, the fast path unchanged.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04_s18
mkdir -p "$OUT"
cd "$REPO"
echo "== parity"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference.py tests/test_gpu_baseline_configs.py -m gpu -q -k "proj or hole or pan or config3 or fill or unusual or far or stream" 2>&1 | tail -30 | tee "$OUT/pytest.log"
echo "== stress"; timeout 600 python tools/stress_projection.py 40 2>&1 | tail -3 | tee "$OUT/stress.log"
echo "== motion sweep"
timeout 300 python tools/probes/proj_motion_sweep.py 2>&1 | grep -v amdgpu.ids | tee "$OUT/motion_sweep.txt"
echo "== burst timing"
timeout 300 python tools/probes/proj_burst.py 2>&1 | grep -v amdgpu.ids | tee "$OUT/burst.txt"
