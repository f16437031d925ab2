#!/bin/bash
# Round 3, GPU session 3: full suite on the new product backward; C=64 forward on 32x32 tiles; PMC of the headline's stripe arms.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r03_s3
mkdir -p "$OUT"
cd "$REPO"
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== full GPU suite"
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -8 | tee "$OUT/pytest_gpu.log"
cp gpurun_out/parity_errors.json "$OUT/" 2>/dev/null
echo "== C=64 forward: 64x16 tiles (-1), 32x32 tiles (33), 32x32 in stripes of 4 (34), 64x16 stripes of 4 (31)"
timeout 600 python tools/bench_ops.py --only fi_fwd --ctx-only --variants=-1,33,34,31 --json "$OUT/bench_ctx64_arms.json" 2>&1 | grep -v amdgpu.ids | tee "$OUT/bench_ctx64_arms.log"
echo "== PMC: C=64 forward arms"
timeout 900 python tools/pmc_any.py --out "$OUT/pmc_ctx64" --match fi_fwd_tiled_c4n -- --only fi_fwd --ctx-only --variants=-1,33,34 2>&1 | grep -v amdgpu.ids | tail -8 | tee "$OUT/pmc_ctx64.log"
echo "== PMC: headline forward, strips (-1) against stripes of 2 (15) and 4 (16) tile columns per XCD"
timeout 900 python tools/pmc_any.py --out "$OUT/pmc_headline" --match fi_fwd_tiled_fs4 -- --only fi_fwd --headline-only --variants=-1,15,16 2>&1 | grep -v amdgpu.ids | tail -8 | tee "$OUT/pmc_headline.log"
echo "== SQ counters of the new RGB backward"
timeout 700 bash tools/pmc_sq.sh r03_s3/sq_fi_bwd fi_bwd fi_bwd_c3_pk 2>&1 | grep -v amdgpu.ids | tail -30 | tee "$OUT/sq_fi_bwd.log"
echo "== PMC traffic of the bench line's kernel (refreshes profiles/traffic.json)"
timeout 900 python tools/pmc_traffic.py --out "$OUT" 2>&1 | tail -25 | tee "$OUT/pmc_traffic.log"
ls "$OUT"
