#!/bin/bash
# Round 3, GPU session 2: the image-gradient-first order of the RGB backward; bench.py's new line.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r03_s2
mkdir -p "$OUT"
cd "$REPO"
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== new tests first"
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_line.py -m gpu -q -x -k "rgb_arms or rgb_scaling or bench_line" 2>&1 | tail -15 | tee "$OUT/pytest_new.log"
echo "== A/B timings (measurement build): fp64 plane per colour (0) / packed, image first 96x32 (20) / dynamic pitch (22) / image gradient first (23)"
timeout 900 python tools/bench_ops.py --only fi_bwd --bwd-variants 0,20,22,23 --json "$OUT/bench_fi_bwd_arms.json" 2>&1 | grep -v amdgpu.ids | tee "$OUT/bench_fi_bwd_arms.log"
echo "== phase clocks"
for k in fi_bwd_pk1 fi_bwd_pk2; do timeout 300 python tools/trace_kernel.py $k 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/fi_bwd_traces.txt"; done
echo "== bench.py (default run)"
timeout 900 python bench.py 2>&1 | tail -2 | tee "$OUT/bench.log"
ls "$OUT"
