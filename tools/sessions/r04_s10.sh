#!/bin/bash
# Round 4, GPU session 10: proj_fill_pending as one wave per tile with the three table walks advancing together.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04_s10
mkdir -p "$OUT"
cd "$REPO"
echo "== parity"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference.py tests/test_gpu_baseline_configs.py -m gpu -q -k "proj or hole or pan or config3 or fill" 2>&1 | tail -6 | tee "$OUT/pytest.log"
echo "== stress"; timeout 600 python tools/stress_projection.py 40 2>&1 | tail -3 | tee "$OUT/stress.log"
echo "== burst timing"
timeout 300 python tools/probes/proj_burst.py 2>&1 | grep -v amdgpu.ids | tee "$OUT/burst.txt"
cd /tmp && export TMPDIR=/tmp
for kind in smooth iid; do
  echo "== kernel trace of the projection calls, flow=$kind"
  timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/prof_$kind" -o proj -- python "$REPO/tools/probes/proj_calls.py" $kind 60 2>&1 | grep "flow=" | tee -a "$OUT/proj_calls.txt"
  python "$REPO/tools/probes/proj_calls_summary.py" "$OUT/prof_$kind/proj_results.db" 150 | grep "fill=1" | tee -a "$OUT/proj_calls.txt"
  rm -rf "$OUT/prof_$kind"
done
