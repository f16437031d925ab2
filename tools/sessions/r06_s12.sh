#!/bin/bash
# Round 6, GPU session 12: the seeded random-shape sweep over every operator (tests/test_gpu_parity.py, RANDOM_CASES).
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r06_s12
mkdir -p "$OUT"
cd "$REPO"
timeout 1500 python -m pytest tests -q -m gpu -k "random_shapes_every_operator" 2>&1 | tail -30 | tee "$OUT/pytest.log"
