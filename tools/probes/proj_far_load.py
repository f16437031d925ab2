#!/usr/bin/env python
"""tools/probes/proj_far_load.py <scale> [pan] [fill] [depth] -- 30 FlowProjection (or DepthFlowProjection) forward calls on
the benchmark's smooth flow times <scale> plus a rigid pan (pan, -pan/2): a workload for rocprofv3 / tools/pmc_sq.sh
(PMC_CMD) that exercises proj_owner_far and the hole filling behind it."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for _p in (ROOT, os.path.join(ROOT, "memc-net_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)
import torch  # noqa: E402

import my_package._ext.my_lib as L  # noqa: E402
from tools import synth  # noqa: E402

arg = sys.argv[1:] + ["", "", "", ""]
scale, pan, fill, depth = float(arg[0] or 4.0), float(arg[1] or 0.0), int(arg[2] or 0), bool(arg[3])
dev = torch.device("cuda:0")
t = synth.torch_inputs(dev, 32, 3, 720, 1280, flow_kind="smooth", with_depth=True)
f, d = (t["flow"] * scale).contiguous(), t["depth"]
f[:, 0] += pan
f[:, 1] -= pan / 2
cnt, out = f.new_zeros((32, 1, 720, 1280)), torch.zeros_like(f)
for _ in range(30):
    if depth:
        L.DepthFlowProjectionLayer_gpu_forward(f, d, cnt, out, fill)
    else:
        L.FlowProjectionLayer_gpu_forward(f, cnt, out, fill)
torch.cuda.synchronize()
