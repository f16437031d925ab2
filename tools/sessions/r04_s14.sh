#!/bin/bash
# Round 4, GPU session 14: edge cases of the projection forward (negative / zero / NaN depths, NaN / Inf flow).
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "unusual" 2>&1 | tail -40
