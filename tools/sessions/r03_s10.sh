#!/bin/bash
# Round 3, GPU session 10: the C=64 forward against the flow's smoothness (are its LDS bank conflicts the flow's doing?).
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r03_s10
mkdir -p "$OUT"
cd "$REPO"
timeout 600 python tools/bench_ops.py --only fi_fwd --ctx-only --ctx-flows --json "$OUT/bench_ctx64_flows.json" 2>&1 | grep -v amdgpu.ids | tee "$OUT/bench_ctx64_flows.log"
cat > /tmp/sq_video.py <<'PY'
import os, sys, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "memc-net_amd"))
import my_package._ext.my_lib as L
from tools import synth
kind = sys.argv[1]
dev = torch.device("cuda:0")
t = synth.torch_inputs(dev, 8, 64, 720, 1280, flow_kind=kind)
out = torch.zeros_like(t["x"])
for _ in range(40):
    L.FilterInterpolationLayer_gpu_forward(t["x"], t["flow"], t["filt"], out)
torch.cuda.synchronize()
PY
cd /tmp && export TMPDIR=/tmp
for kind in smooth video iid; do
  timeout 300 rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace -d $OUT/sq_$kind -o r -- python /tmp/sq_video.py $kind > $OUT/sq_$kind.log 2>&1
  python $REPO/tools/prof_summary.py pmc $OUT/sq_$kind/r_results.db --match fi_fwd_tiled_c4n --out $OUT/sq_$kind.json > /dev/null
  rm -rf $OUT/sq_$kind
  python - <<PY
import json
rows = json.load(open("$OUT/sq_$kind.json"))
v = {r["counter"]: r["mean_value"] for r in rows}
b = v["SQ_BUSY_CU_CYCLES"]
print("$kind: dur_us %.1f  LDS active %.3f  bank conflict %.3f (%.0f%% of LDS-active)  VALU %.3f" % (rows[0]["mean_duration_us"], v["SQ_LDS_IDX_ACTIVE"]/b, v["SQ_LDS_BANK_CONFLICT"]/b, 100*v["SQ_LDS_BANK_CONFLICT"]/v["SQ_LDS_IDX_ACTIVE"], v["SQ_ACTIVE_INST_VALU"]/b))
PY
done 2>&1 | tee "$OUT/sq_by_flow.log"
ls "$OUT"
