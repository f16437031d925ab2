#!/usr/bin/env python
"""tools/probes/proj_calls.py -- the projection forward calls of BASELINE config 3, N each, for a rocprofv3 kernel trace
(which kernel of a call costs what):  rocprofv3 --kernel-trace --stats -d <dir> -o proj -- python tools/probes/proj_calls.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for _p in (ROOT, os.path.join(ROOT, "memc-net_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)
import torch  # noqa: E402

import my_package._ext.my_lib as L  # noqa: E402
from tools import synth  # noqa: E402

dev = torch.device("cuda:0")
kind = sys.argv[1] if len(sys.argv) > 1 else "smooth"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
t = synth.torch_inputs(dev, 32, 3, 720, 1280, flow_kind=kind, with_depth=True)
f, d = t["flow"], t["depth"]
cnt, out = f.new_zeros((32, 1, 720, 1280)), torch.zeros_like(f)
for _ in range(150):
    L.FlowProjectionLayer_gpu_forward(f, cnt, out, 0)
for _ in range(n):
    L.FlowProjectionLayer_gpu_forward(f, cnt, out, 1)
torch.cuda.synchronize()
for _ in range(n):
    L.DepthFlowProjectionLayer_gpu_forward(f, d, cnt, out, 1)
torch.cuda.synchronize()
holes = int((cnt <= 0).sum().item())
print("flow=%s: %d hole cells of %d (%.2f %%)" % (kind, holes, cnt.numel(), 100.0 * holes / cnt.numel()))
