#!/bin/bash
# Round 6, GPU session 11: the many-channel backward passes at ragged widths (whole quads on the owner kernels, tail columns by
# lanes): tests, then the time of 8 x 64 x 720 x W at W = 1280 / 1279 / 1278 / 1277 for both operators.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r06_s11
mkdir -p "$OUT"
cd "$REPO"
timeout 1500 python -m pytest tests -q -m gpu -x -k "many_channel or documented_kernel_paths or unaligned or ragged or strided or capture_many" 2>&1 | tail -4 | tee "$OUT/pytest.log"
timeout 600 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee "$OUT/many_channel_backward_ragged_widths.txt"
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "memc-net_amd"))
import torch
import my_package._ext.my_lib as L
from tools import synth
dev = torch.device("cuda:0")
print("many-channel backward, 8 x 64 x 720 x W, smooth flow; us per call (median of 7), kernel family")
for W in (1280, 1279, 1278, 1277):
    t = synth.torch_inputs(dev, 8, 64, 720, W, flow_kind="smooth", with_grad=True)
    x, f, k, g = t["x"], t["flow"], t["filt"], t["gout"]
    g1, g2, g3 = torch.zeros_like(x), torch.zeros_like(f), torch.zeros_like(k)
    row = []
    for name, fn in (("FilterInterpolation", lambda: L.FilterInterpolationLayer_gpu_backward(x, f, k, g, g1, g2, g3)),
                     ("InterpolationCh", lambda: L.InterpolationChLayer_gpu_backward(x, f, g, g1, g2))):
        for _ in range(2):
            assert fn() == 0
        ts = []
        for _ in range(7):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fn(); b.record(); b.synchronize()
            ts.append(a.elapsed_time(b) * 1e3)
        ts.sort()
        row.append("%s %9.1f us (%s)" % (name, ts[3], L.last_kernel_path()))
    print("W = %d:  " % W + "   ".join(row))
PY
