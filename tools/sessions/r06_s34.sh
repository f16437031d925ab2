#!/bin/bash
# Round 6, GPU session 34: after the `cancel` term of tests/_parity.py -- the case that tripped (both sweeps), the strided sweep of seed
# 777002 again, and two more fresh-seed sweeps (3000 shapes, 1000 strided).
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r06_s34
mkdir -p "$OUT"
cd "$REPO"
MEMC_RANDOM_SEED=777002 MEMC_RANDOM_CASES=600 MEMC_STRIDED_CASES=600 timeout 2400 python -m pytest tests -q -m gpu -k "random_strided_views or (random_shapes_every_operator and 3x9x90x130-smooth-40)" -p no:cacheprovider 2>&1 | tail -4 | tee "$OUT/pytest_777002.log"
MEMC_RANDOM_SEED=777003 MEMC_RANDOM_CASES=3000 timeout 2400 python -m pytest tests -q -m gpu -k "random_shapes_every_operator" -p no:cacheprovider 2>&1 | tail -8 | tee "$OUT/pytest_shapes_777003.log"
MEMC_RANDOM_SEED=777004 MEMC_RANDOM_CASES=1000 MEMC_STRIDED_CASES=1000 timeout 2400 python -m pytest tests -q -m gpu -k "random_strided_views" -p no:cacheprovider 2>&1 | tail -8 | tee "$OUT/pytest_strided_777004.log"
