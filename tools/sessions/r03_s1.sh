#!/bin/bash
# Round 3, GPU session 1: parity of the new RGB backward arms, their timings and phase clocks.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r03_s1
mkdir -p "$OUT"
cd "$REPO"
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== new tests first"
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "rgb_arms or rgb_scaling" 2>&1 | tail -15 | tee "$OUT/pytest_new.log"
echo "== A/B timings (measurement build): fp64 plane per colour (0) / packed aliasing (20) / packed beside (21)"
timeout 900 python tools/bench_ops.py --only fi_bwd --bwd-variants 0,20,21 --json "$OUT/bench_fi_bwd_arms.json" 2>&1 | grep -v amdgpu.ids | tee "$OUT/bench_fi_bwd_arms.log"
echo "== phase clocks"
for k in fi_bwd fi_bwd_pk_alias fi_bwd_pk; do timeout 300 python tools/trace_kernel.py $k 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/fi_bwd_traces.txt"; done
echo "== full GPU suite"
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -15 | tee "$OUT/pytest_gpu.log"
cp gpurun_out/parity_errors.json "$OUT/" 2>/dev/null
ls "$OUT"
