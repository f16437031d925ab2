#!/bin/bash
# Round 4, GPU session 12: SQ counters (incl. the scalar pipe) of the 2 x 2-footprint gather kernels -- are they issue-bound like the
# projection's owner kernel was?  + the parity of the owner kernel's own tile walk (arm -43).
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04_s12
mkdir -p "$OUT"
cd "$REPO"
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "needs_no_zero_fill or pans" 2>&1 | tail -3
bash tools/pmc_sq.sh r04_s12/sq_bl interp "bl_fwd_tiled<3" 2>&1 | tail -34 | tee "$OUT/bl_fwd_sq.txt"
bash tools/pmc_sq.sh r04_s12/sq_pb proj "proj_bwd_tiled<false" 2>&1 | tail -34 | tee "$OUT/proj_bwd_sq.txt"
bash tools/pmc_sq.sh r04_s12/sq_fi fi_fwd "fi_fwd_tiled_fs4" "--headline-only" 2>&1 | tail -34 | tee "$OUT/fi_fwd_sq.txt"
