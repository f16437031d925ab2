#!/usr/bin/env python
"""tools/probes/fi_phase_sweep.py -- the RGB adaptive warp forward with PHASED STORES (measurement arm 26): every workgroup holds its
results until the chip-wide write window of the 100 MHz clock opens.  One process; the product kernel between every two settings.
    python tools/probes/fi_phase_sweep.py [smooth|iid] [rounds]"""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for _p in (ROOT, os.path.join(ROOT, "memc-net_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)
import torch  # noqa: E402

from tools import measure as M  # noqa: E402
from tools import synth  # noqa: E402

flow = sys.argv[1] if len(sys.argv) > 1 else "smooth"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
L = M.bound()
dev = torch.device("cuda:0")
t = synth.torch_inputs(dev, 32, 3, 720, 1280, flow_kind=flow)
x, f, k = t["x"], t["flow"], t["filt"]
out = torch.zeros_like(x)
want = torch.zeros_like(x)
M.set_variant("fi_fwd", -1)
L.FilterInterpolationLayer_gpu_forward(x, f, k, want)
cases = [None] + [(per, win) for per in (500, 800, 1000, 1200, 1500, 2000) for win in (per // 10, per // 8, per // 6, per // 4)]
ts = {c: [] for c in cases}
for _ in range(200):
    L.FilterInterpolationLayer_gpu_forward(x, f, k, out)
for r in range(rounds):
    for c in cases:
        if c is None:
            M.set_variant("fi_fwd", -1)
        else:
            M.set_variant("fi_phase", c[0] * 65536 + c[1])
            M.set_variant("fi_fwd", 26)
        for _ in range(3):
            L.FilterInterpolationLayer_gpu_forward(x, f, k, out)
        for _ in range(8):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); L.FilterInterpolationLayer_gpu_forward(x, f, k, out); b.record(); b.synchronize()
            ts[c].append(a.elapsed_time(b) * 1e3)
        if r == 0 and c is not None:
            assert torch.equal(out, want), "phased stores changed the result"
M.set_variant("fi_fwd", -1)
base = statistics.median(ts[None])
for c in cases:
    m = statistics.median(ts[c])
    print("flow=%-6s %-44s %8.1f us (%.3f)" % (flow, "product" if c is None else "stores in the last %.2f us of every %.1f us" % (
        c[1] / 100.0, c[0] / 100.0), m, m / base), flush=True)
