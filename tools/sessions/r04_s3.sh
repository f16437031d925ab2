#!/bin/bash
# Round 4, GPU session 3: the fill epilogue with a dealt hole list + pending flag; rcp in the read-out; the heavy-tailed tile test
# in full; the RGB backward against the previous build in one process.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04_s3
mkdir -p "$OUT"
cd "$REPO"
echo "== parity: projection + RGB backward"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference.py tests/test_gpu_baseline_configs.py -m gpu -q -k "proj or hole or pan or config3 or fill or backward or bwd or heavy or special or config2 or headline" 2>&1 | tail -40 | tee "$OUT/pytest.log"
echo "== stress"; timeout 600 python tools/stress_projection.py 40 2>&1 | tail -6 | tee "$OUT/stress.log"
echo "== projection A/B in one process (-1 product, -42 without proj_fill_pending, -40 round 3's set)"
timeout 600 python tools/ab_variants.py --op projection --variants=-1,-42,-40 --cases proj,proj_fill,depth_fill --flows smooth,iid 2>&1 | grep -v amdgpu.ids | tee "$OUT/ab_fill.txt"
if [ -f memc-net_amd/lib/libmemc_hip_prev.so ]; then
  echo "== RGB backward: previous build (A) against this one (B), one process"
  timeout 600 python tools/ab_libs.py memc-net_amd/lib/libmemc_hip_prev.so memc-net_amd/lib/libmemc_hip.so --op fi_bwd,fi_fwd 2>&1 | grep -v amdgpu.ids | tee "$OUT/ab_bwd_libs.txt"
fi
cd /tmp && export TMPDIR=/tmp
for kind in smooth iid; do
  echo "== kernel trace of the projection calls, flow=$kind"
  timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/prof_$kind" -o proj -- python "$REPO/tools/probes/proj_calls.py" $kind 60 2>&1 | grep "flow=" | tee -a "$OUT/proj_calls.txt"
  python "$REPO/tools/probes/proj_calls_summary.py" "$OUT/prof_$kind/proj_results.db" 150 | tee -a "$OUT/proj_calls.txt"
  rm -rf "$OUT/prof_$kind"
done
