#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r02i
mkdir -p "$OUT"
cd "$REPO"
echo "== projection tests"; timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference.py tests/test_gpu_baseline_configs.py -m gpu -q -k "projection or hole or stream or capture or config3 or reference_kernels or launcher" 2>&1 | tail -5 | tee "$OUT/pytest_proj.log"
echo "== stress"; timeout 600 python tools/stress_projection.py 45 > "$OUT/stress.log" 2>&1; grep -o "MISMATCH variant=[-0-9]*" "$OUT/stress.log" | sort | uniq -c; tail -1 "$OUT/stress.log"
echo "== sweep"
timeout 900 python tools/bench_ops.py --only proj --quick --proj-variants=114 --json "$OUT/bench_proj.json" 2>&1 | tee "$OUT/bench_proj.log" | grep -v "^$" | cut -c1-150
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/prof_proj" -o proj -- python "$REPO/tools/bench_ops.py" --only proj --quick --json "$OUT/bench_proj_prof.json" > "$OUT/prof_proj.log" 2>&1
python "$REPO/tools/prof_summary.py" stats "$OUT/prof_proj/proj_results.db" --out "$OUT/proj_kernel_stats.txt" | grep -v "^at::\|^$" | head -12
rm -rf "$OUT/prof_proj"
