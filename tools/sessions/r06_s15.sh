#!/bin/bash
# Round 6, GPU session 15: where the many-channel backward's time goes at a ragged width (kernel trace, 8 x 64 x 720 x 1278 against 1280).
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r06_s15
mkdir -p "$OUT"
cd "$REPO"
cat > /tmp/rag.py <<'PY'
import os, sys
R = os.environ["MEMC_REPO"]
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "memc-net_amd"))
import torch
import my_package._ext.my_lib as L
from tools import synth
dev = torch.device("cuda:0")
W = int(sys.argv[1])
t = synth.torch_inputs(dev, 8, 64, 720, W, flow_kind="smooth", with_grad=True)
x, f, k, g = t["x"], t["flow"], t["filt"], t["gout"]
g1, g2, g3 = torch.zeros_like(x), torch.zeros_like(f), torch.zeros_like(k)
for _ in range(6):
    L.FilterInterpolationLayer_gpu_backward(x, f, k, g, g1, g2, g3)
    L.InterpolationChLayer_gpu_backward(x, f, g, g1, g2)
torch.cuda.synchronize()
PY
cd /tmp && export TMPDIR=/tmp MEMC_REPO=$REPO
for W in 1280 1278; do
  timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/prof_$W" -o rag -- python /tmp/rag.py $W > "$OUT/prof_$W.log" 2>&1
  echo "== W = $W" | tee -a "$OUT/many_channel_ragged_kernel_trace.txt"
  python "$REPO/tools/prof_summary.py" stats "$OUT/prof_$W/rag_results.db" | grep "memc::" | tee -a "$OUT/many_channel_ragged_kernel_trace.txt"
  rm -rf "$OUT/prof_$W"
done
