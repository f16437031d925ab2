#!/bin/bash
# Round 6, GPU session 35: the strided sweeps of seeds 777002 / 777004 again (the backward now gets the same forward planes on both sides).
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r06_s35
mkdir -p "$OUT"
cd "$REPO"
MEMC_RANDOM_SEED=777002 MEMC_RANDOM_CASES=600 MEMC_STRIDED_CASES=600 timeout 2400 python -m pytest tests -q -m gpu -k "random_strided_views" -p no:cacheprovider 2>&1 | tail -4 | tee "$OUT/pytest_strided_777002.log"
MEMC_RANDOM_SEED=777004 MEMC_RANDOM_CASES=1000 MEMC_STRIDED_CASES=1000 timeout 2400 python -m pytest tests -q -m gpu -k "random_strided_views" -p no:cacheprovider 2>&1 | tail -8 | tee "$OUT/pytest_strided_777004.log"
