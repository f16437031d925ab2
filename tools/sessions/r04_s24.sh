#!/bin/bash
# Round 4, GPU session 24: proj_fill_pending walks beyond the tile once per row side / column (not per hole), four holes per
# lane in flight -- parity, stress, small pans (hole bands), the ordinary case's kernel trace.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04_s24
mkdir -p "$OUT"
cd "$REPO"
echo "== parity"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference.py tests/test_gpu_baseline_configs.py -m gpu -q -k "proj or hole or pan or config3 or fill or unusual or far or stream" 2>&1 | tail -5 | tee "$OUT/pytest.log"
echo "== stress"; timeout 600 python tools/stress_projection.py 40 2>&1 | tail -1 | tee "$OUT/stress.log"
echo "== small pans"
timeout 300 python tools/probes/proj_small_pans.py 2>&1 | grep -v amdgpu.ids | tee "$OUT/small_pans.txt"
echo "== motion sweep, all operators"
timeout 300 python tools/probes/motion_sweep_all.py 2>&1 | grep -v amdgpu.ids | grep -A7 panned | tee "$OUT/motion_sweep_pans.txt"
echo "== burst timing"
timeout 300 python tools/probes/proj_burst.py 2>&1 | grep -v amdgpu.ids | tee "$OUT/burst.txt"
echo "== projection kernels of a call (kernel trace)"
( cd /tmp && export TMPDIR=/tmp && for kind in smooth iid; do
  timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/prof_proj_$kind" -o proj -- python "$REPO/tools/probes/proj_calls.py" $kind 60 2>&1 | grep "flow=" | tee -a "$OUT/proj_calls.txt"
  python "$REPO/tools/probes/proj_calls_summary.py" "$OUT/prof_proj_$kind/proj_results.db" 150 | grep "pending" | tee -a "$OUT/proj_calls.txt"
  rm -rf "$OUT/prof_proj_$kind"; done )
