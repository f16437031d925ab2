#!/bin/bash
# Round 6, GPU session 14: the random sweep on strided views (every tensor a window of a larger buffer), 300 cases once.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r06_s14
mkdir -p "$OUT"
cd "$REPO"
sed -i 's/RANDOM_CASES\[:48\]/RANDOM_CASES[:300]/g' tests/test_gpu_parity.py      # (in the box's copy only)
MEMC_RANDOM_CASES=300 timeout 2400 python -m pytest tests -q -m gpu -k "random_strided_views" 2>&1 | tail -40 | tee "$OUT/pytest.log"
