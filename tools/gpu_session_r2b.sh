#!/bin/bash
# Round 2, session B: where does proj_owner2's time go?  timing arms + phase timestamps
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r02b
mkdir -p "$OUT"
cd "$REPO"
echo "== arms (200 + 10 * arm + log2(TH/16)): 0 owner alone, 1 no scan, 2 no adds, 3 no read-out, 4 no halo, 5 no loads no scan"
timeout 900 python tools/bench_ops.py --only proj --quick --proj-variants=200,210,220,230,240,250,201,211,221,231,241,251 --json "$OUT/bench_arms.json" 2>&1 | grep "fillhole=0" | tee "$OUT/bench_arms.log"
echo "== timestamps"
for k in proj proj2_16 proj2_32; do timeout 300 python tools/trace_kernel.py $k 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/trace.log"; done
ls "$OUT"
