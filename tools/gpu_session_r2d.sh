#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r02d
mkdir -p "$OUT"
cd "$REPO"
echo "== occupancy arms: 26x = two planes only (3 WG/CU at TH=32)"
timeout 900 python tools/bench_ops.py --only proj --quick --proj-variants=200,260,201,261,140 --json "$OUT/bench_arms.json" 2>&1 | grep "fillhole=0" | tee "$OUT/bench_arms.log"
echo "== projection tests"; timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference.py tests/test_gpu_baseline_configs.py -m gpu -q -k "projection or hole or stream or capture or config3 or reference_kernels or launcher" 2>&1 | tail -12 | tee "$OUT/pytest_proj.log"
echo "== stress"; timeout 600 python tools/stress_projection.py 45 > "$OUT/stress.log" 2>&1; grep -o "MISMATCH variant=[-0-9]*" "$OUT/stress.log" | sort | uniq -c; tail -1 "$OUT/stress.log"
