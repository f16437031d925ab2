#!/bin/bash
# Round 4, GPU session 5: the projection's scratch block cached per stream (host enqueue time), parity of everything touched.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04_s5
mkdir -p "$OUT"
cd "$REPO"
echo "== parity"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference.py tests/test_gpu_baseline_configs.py -m gpu -q -k "proj or hole or pan or config3 or fill or backward or bwd or heavy or special or config2 or headline" 2>&1 | tail -30 | tee "$OUT/pytest.log"
echo "== stress"; timeout 600 python tools/stress_projection.py 30 2>&1 | tail -3 | tee "$OUT/stress.log"
echo "== burst timing"
timeout 300 python tools/probes/proj_burst.py 2>&1 | grep -v amdgpu.ids | tee "$OUT/burst.txt"
echo "== projection A/B in one process"
timeout 600 python tools/ab_variants.py --op projection --variants=-1,-40 --cases proj,proj_fill,depth,depth_fill --flows smooth,iid 2>&1 | grep -v amdgpu.ids | tee "$OUT/ab_fill.txt"
