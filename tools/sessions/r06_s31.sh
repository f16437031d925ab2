#!/bin/bash
# Round 6, GPU session 31: the random-shape sweeps of the parity suite on the final build with FRESH seeds: 1500 shapes through every
# operator, 600 of them again as strided views.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r06_s31
mkdir -p "$OUT"
cd "$REPO"
MEMC_RANDOM_SEED=777001 MEMC_RANDOM_CASES=1500 timeout 2400 python -m pytest tests -q -m gpu -x -k "random_shapes_every_operator" -p no:cacheprovider 2>&1 | tail -6 | tee "$OUT/pytest_shapes.log"
MEMC_RANDOM_SEED=777002 MEMC_RANDOM_CASES=600 MEMC_STRIDED_CASES=600 timeout 2400 python -m pytest tests -q -m gpu -x -k "random_strided_views" -p no:cacheprovider 2>&1 | tail -6 | tee "$OUT/pytest_strided.log"
