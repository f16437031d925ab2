#!/bin/bash
# Round 3, GPU session 23: SQ counters of the column-mapped arm (40) against the product.
# (Arm 40 was removed after sessions 22-24: the script is the record of what ran, it no longer selects that kernel.)
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r03_s23
mkdir -p "$OUT"
cd "$REPO"
cat > /tmp/sq_arm.py <<'PY'
import os, sys, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "memc-net_amd"))
from tools import measure as M
from tools import synth
L = M.bound()
dev = torch.device("cuda:0")
t = synth.torch_inputs(dev, 8, 64, 720, 1280, flow_kind="smooth")
out = torch.zeros_like(t["x"])
for v in (40, -1):
    M.set_variant("fi_fwd", v)
    for _ in range(20):
        L.FilterInterpolationLayer_gpu_forward(t["x"], t["flow"], t["filt"], out)
torch.cuda.synchronize()
PY
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS --kernel-trace -d $OUT/sq -o r -- python /tmp/sq_arm.py > $OUT/sq.log 2>&1
python $REPO/tools/prof_summary.py pmc $OUT/sq/r_results.db --match fi_fwd --out $OUT/sq.json > /dev/null
rm -rf $OUT/sq
python - <<PY | tee "$OUT/sq_counters.txt"
import json
rows = json.load(open("$OUT/sq.json"))
for k in sorted(set(r["kernel"] for r in rows)):
    v = {r["counter"]: r["mean_value"] for r in rows if r["kernel"] == k}
    d = [r for r in rows if r["kernel"] == k][0]["mean_duration_us"]
    b = v["SQ_BUSY_CU_CYCLES"]
    print("%-50s %7.1f us  LDS active %.3f  conflict %.3f  VALU %.3f  wait_lds %.3f  cycles/LDS inst %.2f" % (k[:50], d, v["SQ_LDS_IDX_ACTIVE"]/b, v["SQ_LDS_BANK_CONFLICT"]/b, v["SQ_ACTIVE_INST_VALU"]/b, v["SQ_WAIT_INST_LDS"]/b, v["SQ_LDS_IDX_ACTIVE"]/v["SQ_INSTS_LDS"]))
PY
