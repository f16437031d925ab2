#!/bin/bash
# Round 3, GPU session 21: the LDS probe with planar-layout gathers (patterns 11-19) (tools/probes/lds_conflict_probe.hip).
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r03_s21
mkdir -p "$OUT"
cd "$REPO"
python tools/probes/lds_conflict_probe.py 2>&1 | grep -v amdgpu.ids | tee "$OUT/probe_times.log"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CU_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --kernel-trace -d $OUT/p -o r -- python $REPO/tools/probes/lds_conflict_probe.py > $OUT/p.log 2>&1
python $REPO/tools/prof_summary.py pmc $OUT/p/r_results.db --match lds_probe --out $OUT/pmc.json > /dev/null
rm -rf $OUT/p
python - <<PY | tee "$OUT/lds_counter_probe.txt"
import json
rows = json.load(open("$OUT/pmc.json"))
ker = sorted(set(r["kernel"] for r in rows))
for k in ker:
    v = {r["counter"]: r["mean_value"] for r in rows if r["kernel"] == k}
    d = [r for r in rows if r["kernel"] == k][0]["mean_duration_us"]
    print("%-40s %7.1f us  LDS_IDX_ACTIVE %.3e  BANK_CONFLICT %.3e (%.0f%%)  INSTS_LDS %.3e  cycles/inst %.2f" % (
        k[:40], d, v["SQ_LDS_IDX_ACTIVE"], v["SQ_LDS_BANK_CONFLICT"], 100 * v["SQ_LDS_BANK_CONFLICT"] / v["SQ_LDS_IDX_ACTIVE"],
        v["SQ_INSTS_LDS"], v["SQ_LDS_IDX_ACTIVE"] / v["SQ_INSTS_LDS"]))
PY
