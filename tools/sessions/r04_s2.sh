#!/bin/bash
# Round 4, GPU session 2: (1) the RGB backward passes with per-site block exponents: parity incl. the heavy-tailed tile, A/B of the
# kernel time against the committed build; (2) which kernel of a projection call with hole filling costs what (kernel trace), the
# call without proj_fill_pending (arm -42).
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04_s2
mkdir -p "$OUT"
cd "$REPO"
echo "== parity: RGB backward passes"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference.py tests/test_gpu_baseline_configs.py -m gpu -q -x -k "backward or bwd or heavy or special or config2 or headline" 2>&1 | tail -8 | tee "$OUT/pytest_bwd.log"
echo "== RGB backward timings (product library)"
timeout 600 python tools/bench_ops.py --only fi_bwd,interp --json "$OUT/bench_bwd.json" 2>&1 | grep -v amdgpu.ids | tee "$OUT/bench_bwd.log"
echo "== projection: fill arms in one process (-1 product, -42 without proj_fill_pending)"
timeout 600 python tools/ab_variants.py --op projection --variants=-1,-42 --cases proj,proj_fill,depth_fill --flows smooth,iid 2>&1 | grep -v amdgpu.ids | tee "$OUT/ab_fill.txt"
cd /tmp && export TMPDIR=/tmp
for kind in smooth iid; do
  echo "== kernel trace of the projection calls, flow=$kind"
  timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/prof_$kind" -o proj -- python "$REPO/tools/probes/proj_calls.py" $kind 60 2>&1 | grep "flow=" | tee -a "$OUT/proj_calls.txt"
  python "$REPO/tools/probes/proj_calls_summary.py" "$OUT/prof_$kind/proj_results.db" 150 | tee -a "$OUT/proj_calls.txt"
  rm -rf "$OUT/prof_$kind"
done
ls "$OUT"
