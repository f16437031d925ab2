#!/usr/bin/env python
"""tools/probes/chk_fi_fwd_variants.py -- measurement-build arms of the RGB adaptive warp forward against the PRODUCT library:
results (max abs difference, must be 0: same arithmetic, another tile shape) on the benchmark's flows.
    python tools/probes/chk_fi_fwd_variants.py 20,21,22,23,24,25"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for _p in (ROOT, os.path.join(ROOT, "memc-net_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)
import torch  # noqa: E402

import my_package._ext.my_lib as P  # noqa: E402  (the product library)
from tools import measure as M  # noqa: E402
from tools import synth  # noqa: E402

variants = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "20,21,22,23,24,25").split(",")]
L = M.bound()
dev = torch.device("cuda:0")
for shape in ((4, 3, 720, 1280), (2, 3, 256, 448), (1, 3, 100, 260)):
    for flow in ("smooth", "iid"):
        for scale in (1.0, 3.0):
            t = synth.torch_inputs(dev, *shape, flow_kind=flow, seed=11)
            f = (t["flow"] * scale).contiguous()
            want = torch.full_like(t["x"], float("nan"))
            assert P.FilterInterpolationLayer_gpu_forward(t["x"], f, t["filt"], want) == 0
            for v in variants:
                M.set_variant("fi_fwd", v)
                got = torch.full_like(t["x"], float("nan"))
                rc = L.FilterInterpolationLayer_gpu_forward(t["x"], f, t["filt"], got)
                torch.cuda.synchronize()
                d = (got - want).abs().max().item() if rc == 0 else float("nan")
                print("shape %-18s flow %-6s x%.0f variant %3d rc %d max|diff| %g %s" % (
                    shape, flow, scale, v, rc, d, "" if d == 0 else "<-- DIFFERS"), flush=True)
            M.set_variant("fi_fwd", -1)
