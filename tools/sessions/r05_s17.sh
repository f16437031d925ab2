#!/bin/bash
# Round 5, GPU session 17: the last two operators at ragged widths (FilterInterpolation forward, the projections' backward) on
# their tiled kernels: the new parity test, the kernel-path test, every test that touches those two operators, the slow-path table.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05_s17
mkdir -p "$OUT"
cd "$REPO"
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py tests/test_gpu_reference.py -q -m gpu -k "ragged or paths or unaligned or multiples or forward or Forward or backward or Backward or fi_ or proj" 2>&1 | tail -5 | tee $OUT/pytest.log
timeout 400 python tools/probes/slow_paths.py 2>&1 | grep -v amdgpu.ids | tee $OUT/slow_paths.txt
