#!/bin/bash
# Round 6, GPU session 13: the random-shape sweep at 600 cases (the suite runs 64).
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r06_s13
mkdir -p "$OUT"
cd "$REPO"
MEMC_RANDOM_CASES=600 timeout 2400 python -m pytest tests -q -m gpu -k "random_shapes_every_operator" 2>&1 | tail -40 | tee "$OUT/pytest.log"
