#!/bin/bash
# Round 6, GPU session 7 (round-5 review, item 5): the fill epilogue where nearly every tile holds holes.  Measurement arms of
# proj_owner5 / owner_fill_epilogue (PENDT): tiles with more than T lanes holding a hole skip the in-tile fill and leave ALL their
# holes to proj_fill_pending -- -51: T = 0 (every tile with a hole), -52: T = 8, -53: T = 32 -- against the product (-1), one
# process; on the benchmark's flow, the flow twice as large, and under camera pans of 40 and 8 px.  First: same results?
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r06_s7
mkdir -p "$OUT"
cd "$REPO"
timeout 300 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee "$OUT/pendt_same_results.txt"
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "memc-net_amd"))
import torch
from tools import measure as M, synth
M.use(); L = M.bound(); dev = torch.device("cuda:0")
t = synth.torch_inputs(dev, 8, 3, 720, 1280, flow_kind="smooth", with_depth=True)
for name, f in (("benchmark flow", t["flow"]), ("flow x 2", t["flow"] * 2), ("pan 40", t["flow"] + torch.tensor([40.0, -20.0], device=dev).view(1, 2, 1, 1))):
    f = f.contiguous(); res = {}
    for v in (-1, -51, -52, -53):
        M.set_variant("projection", v)
        c, o = torch.full((8, 1, 720, 1280), 7.0, device=dev), torch.full((8, 2, 720, 1280), 7.0, device=dev)
        assert L.FlowProjectionLayer_gpu_forward(f, c, o, 1) == 0
        dc, do = torch.full((8, 1, 720, 1280), 7.0, device=dev), torch.full((8, 2, 720, 1280), 7.0, device=dev)
        assert L.DepthFlowProjectionLayer_gpu_forward(f, t["depth"], dc, do, 1) == 0
        res[v] = (c, o, dc, do)
    M.set_variant("projection", -1)
    for v in (-51, -52, -53):
        print("%-15s variant %d: count equal %s, max |out diff| %.3g; depth: max |count diff| %.3g, max |out diff| %.3g" % (
            name, v, bool(torch.equal(res[v][0], res[-1][0])), float((res[v][1] - res[-1][1]).abs().max()),
            float((res[v][2] - res[-1][2]).abs().max()), float((res[v][3] - res[-1][3]).abs().max())))
PY
for extra in "" "--scale 2" "--pan 40" "--pan 8" "--scale 1.5"; do
  timeout 600 python tools/ab_variants.py --op projection --variants=-1,-51,-52,-53 --cases proj_fill,depth_fill --rounds 6 $extra 2>&1 | grep -v amdgpu.ids | sed "s/^/[$extra] /" | tee -a "$OUT/pendt_ab.txt"
done
