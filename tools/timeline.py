#!/usr/bin/env python
"""Per-launch duration time series of the headline kernel (warm-up / DVFS behaviour)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "memc-net_amd")):
    sys.path.insert(0, p)
import torch
import my_package._ext.my_lib as L
from tools import measure as M  # noqa: E402
M.use()                             # the measurement build: ablation / A-B arms live only there
from tools import synth
dev = torch.device("cuda:0")
t = synth.torch_inputs(dev, 32, 3, 720, 1280)
x, f, k = t["x"], t["flow"], t["filt"]; out = torch.zeros_like(x)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1200
variants = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [-1]
for v in variants:
    M.set_variant("fi_fwd", v)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    torch.cuda.synchronize()
    for a, b in ev:
        a.record(); L.FilterInterpolationLayer_gpu_forward(x, f, k, out); b.record()
    torch.cuda.synchronize()
    d = [a.elapsed_time(b) * 1e3 for a, b in ev]
    chunks = [sum(d[i:i + 100]) / 100 for i in range(0, n, 100)]
    print("variant", v, "mean us per 100 launches:", " ".join("%.0f" % c for c in chunks), flush=True)
