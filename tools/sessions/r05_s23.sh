#!/bin/bash
# Round 5, GPU session 23: the whole -m gpu suite and smoke() on the round's final build (the full session r05b ran before the
# last two source changes: FilterInterpolation forward at many channels and ragged widths, the dominant motion's dead zone).
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05_s23
mkdir -p "$OUT"
cd "$REPO"
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee "$OUT/smoke.log"
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -6 | tee "$OUT/pytest_gpu.log"
cp gpurun_out/parity_errors.json "$OUT/parity_errors.json" 2>/dev/null
timeout 600 python tools/bench_ops.py --quick --json "$OUT/bench_ops.json" 2>&1 | grep -v amdgpu.ids | tee "$OUT/bench_ops.log"
