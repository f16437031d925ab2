#!/bin/bash
# Round 6, GPU session 10: the final build -- smoke, the whole GPU suite (with the record of observed errors), the bench line at the
# driver's arguments, the one-rank RCCL communicator's record.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r06_s10
mkdir -p "$OUT"
cd "$REPO"
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee "$OUT/smoke.log"
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -6 | tee "$OUT/pytest_gpu.log"
cp gpurun_out/parity_errors.json "$OUT/parity_errors.json" 2>/dev/null
T0=$(date +%s.%N)
timeout 900 python bench.py --steps 20 --warmup 5 > "$OUT/bench_driver_args.log" 2>&1
python -c "import sys; print('bench.py --steps 20 --warmup 5 took %.1f s end to end' % (float(sys.argv[2]) - float(sys.argv[1])))" $T0 $(date +%s.%N) | tee "$OUT/bench_wall.txt"
tail -1 "$OUT/bench_driver_args.log" | cut -c1-300
timeout 600 python bench.py --dist-single --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>&1 | grep '^{"metric"' > "$OUT/bench_dist_single.log"
python -c "
import json; d=json.loads(open('$OUT/bench_dist_single.log').read()); print(json.dumps(d['dist']))"
