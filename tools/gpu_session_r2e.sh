#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r02e
mkdir -p "$OUT"
cd "$REPO"
echo "== projection tests"; timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference.py tests/test_gpu_baseline_configs.py -m gpu -q -k "projection or hole or stream or capture or config3 or reference_kernels or launcher" 2>&1 | tail -12 | tee "$OUT/pytest_proj.log"
echo "== stress"; timeout 600 python tools/stress_projection.py 45 > "$OUT/stress.log" 2>&1; grep -o "MISMATCH variant=[-0-9]*" "$OUT/stress.log" | sort | uniq -c; tail -1 "$OUT/stress.log"
echo "== sweep"
timeout 900 python tools/bench_ops.py --only proj --proj-variants=-10,100,104,110,112,114,120,140,164 --json "$OUT/bench_proj.json" 2>&1 | tee "$OUT/bench_proj.log" | grep -v "^$" | cut -c1-150
