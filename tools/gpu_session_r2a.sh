#!/bin/bash
# Round 2, session A: correctness of the new projection forward / measurement split, projection geometry sweep.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r02a
mkdir -p "$OUT"
cd "$REPO"
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== smoke";   timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4 | tee "$OUT/smoke.log"
echo "== projection tests first"; timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "projection or hole or stream or capture" 2>&1 | tail -15 | tee "$OUT/pytest_proj.log"
echo "== stress"; timeout 600 python tools/stress_projection.py 45 2>&1 | tail -12 | tee "$OUT/stress.log"
echo "== projection sweep"; timeout 900 python tools/bench_ops.py --only proj --proj-variants=-10,100,102,104,110,112,114,120,122,124 --json "$OUT/bench_proj.json" 2>&1 | tee "$OUT/bench_proj.log"
echo "== pytest all"; timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -25 | tee "$OUT/pytest_gpu.log"
echo "== bench"; timeout 600 python bench.py 2>&1 | tail -2 | tee "$OUT/bench.log"
cd /tmp && export TMPDIR=/tmp
echo "== rocprof kernel trace of the projection rows (product library)"
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/prof_proj" -o proj -- python "$REPO/tools/bench_ops.py" --only proj --quick --json "$OUT/bench_proj_prof.json" > "$OUT/prof_proj.log" 2>&1
python "$REPO/tools/prof_summary.py" stats "$OUT/prof_proj/proj_results.db" --out "$OUT/proj_kernel_stats.txt" | grep -v "^at::\|^$" | head -30
rm -rf "$OUT/prof_proj"
ls "$OUT"
