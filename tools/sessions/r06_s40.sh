#!/bin/bash
# Round 6, GPU session 40: the final build of the round -- smoke, the whole -m gpu suite, the bench line at the driver's arguments and at
# the defaults (with the padded-layout secondary row).
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r06_s40
mkdir -p "$OUT"
cd "$REPO"
export HSA_ENABLE_IPC_MODE_LEGACY=0
sha256sum memc-net_amd/lib/libmemc_hip.so | tee "$OUT/lib_sha256.txt"
echo "== smoke";   timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee "$OUT/smoke.log"
echo "== pytest";  timeout 1800 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 | tee "$OUT/pytest_gpu.log"
cp gpurun_out/parity_errors.json "$OUT/parity_errors.json" 2>/dev/null
echo "== bench at the driver's arguments"
timeout 900 python bench.py --steps 20 --warmup 5 2>&1 | tail -1 | tee "$OUT/bench_driver_args.json" | cut -c1-600
echo "== bench, defaults"
timeout 900 python bench.py 2>&1 | tail -1 | tee "$OUT/bench_line.json" | cut -c1-600
