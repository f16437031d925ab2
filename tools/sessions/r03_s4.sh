#!/bin/bash
# Round 3, GPU session 4: full suite on the cleaned sources + the packed-plane bilinear backward; its A/B; the model's dense-layer switches.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r03_s4
mkdir -p "$OUT"
cd "$REPO"
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== full GPU suite"
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -12 | tee "$OUT/pytest_gpu.log"
cp gpurun_out/parity_errors.json "$OUT/" 2>/dev/null
echo "== RGB bilinear backward A/B"
timeout 600 python tools/ab_bl_bwd.py 2>&1 | grep -v amdgpu.ids | tee "$OUT/ab_bl_bwd.log"
echo "== operator sweep (product library)"
timeout 900 python tools/bench_ops.py --json "$OUT/bench_ops.json" 2>&1 | grep -v amdgpu.ids | tee "$OUT/bench_ops.log"
echo "== model: default / MIOpen solver search / channels-last / both"
timeout 600 python tools/bench_model.py --json "$OUT/bench_model.json" 2>&1 | tail -1 | tee "$OUT/bench_model.log"
timeout 900 python tools/bench_model.py --miopen-search --json "$OUT/bench_model_search.json" 2>&1 | tail -1 | tee -a "$OUT/bench_model.log"
timeout 600 python tools/bench_model.py --channels-last --json "$OUT/bench_model_nhwc.json" 2>&1 | tail -1 | tee -a "$OUT/bench_model.log"
timeout 900 python tools/bench_model.py --miopen-search --channels-last --json "$OUT/bench_model_search_nhwc.json" 2>&1 | tail -1 | tee -a "$OUT/bench_model.log"
ls "$OUT"
