#!/bin/bash
# Round 4, GPU session 7: the RGB backward as two halves (arm 62) for small grids and without the image gradient (gradinput1 NULL).
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04_s7
mkdir -p "$OUT"
cd "$REPO"
echo "== parity"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ve_call_trace.py -m gpu -q -k "backward or bwd or heavy or special or image_gradient or trace or calls" 2>&1 | tail -12 | tee "$OUT/pytest.log"
echo "== RGB backward: one kernel (60) against its two halves (62), and without the image gradient; one process"
timeout 600 python tools/ab_variants.py --op fi_bwd --variants=60,62 --cases fi_bwd_c2,fi_bwd,fi_bwd_c2_nog1,fi_bwd_nog1 --flows smooth,iid 2>&1 | grep -v amdgpu.ids | tee "$OUT/ab_fi_bwd_halves.txt"
