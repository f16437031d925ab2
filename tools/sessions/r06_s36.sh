#!/bin/bash
# Round 6, GPU session 36: the random sweeps on LARGER shapes (MEMC_RANDOM_BIG=1: any height up to 420, any width up to 900, batch up to 5):
# 500 through every operator, 250 of them again as strided views.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r06_s36
mkdir -p "$OUT"
cd "$REPO"
MEMC_RANDOM_BIG=1 MEMC_RANDOM_SEED=888001 MEMC_RANDOM_CASES=500 MEMC_STRIDED_CASES=250 timeout 3000 python -m pytest tests -q -m gpu -k "random_shapes_every_operator or random_strided_views" -p no:cacheprovider 2>&1 | tail -15 | tee "$OUT/pytest_big.log"
