#!/bin/bash
# Round 6, GPU session 8: the PMC traffic passes the full session r06a lost (the calibration probe's library was not built in
# the fresh tree), then session 7's fill-epilogue arms.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r06a
mkdir -p "$OUT"
cd "$REPO"
timeout 900 python tools/pmc_traffic.py --out "$OUT" 2>&1 | tail -30
timeout 1500 python tools/pmc_traffic_ops.py --out "$OUT" 2>&1 | tail -16
bash tools/sessions/r06_s7.sh
