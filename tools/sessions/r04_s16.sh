#!/bin/bash
# Round 4, GPU session 16: the owner kernel's scan laid out 32 quads wide (coordinates by adds) -- previous build (A) against this one (B).
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04_s16
mkdir -p "$OUT"
cd "$REPO"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference.py tests/test_gpu_baseline_configs.py -m gpu -q -k "proj or hole or pan or config3 or fill or unusual" 2>&1 | tail -4 | tee "$OUT/pytest.log"
timeout 600 python tools/stress_projection.py 40 2>&1 | tail -2 | tee "$OUT/stress.log"
timeout 900 python tools/ab_libs.py memc-net_amd/lib/libmemc_hip_prev.so memc-net_amd/lib/libmemc_hip.so --op proj,proj_fill,depth_fill 2>&1 | grep -v amdgpu.ids | tee "$OUT/ab_scan_layout.txt"
bash tools/pmc_sq.sh r04_s16/sq proj "proj_owner5<false" 2>&1 | grep -E "dur_us|/ SQ_BUSY|/ wave" | tee "$OUT/proj_owner5_sq.txt"
