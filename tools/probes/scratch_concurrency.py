#!/usr/bin/env python
"""tools/probes/scratch_concurrency.py -- does a kernel that SPILLS (private-segment scratch) on one stream disturb another
stream's results?  Round 4 saw exactly that with an experimental build of proj_owner_far (80 VGPRs, 76 B of scratch per
lane): the OTHER stream's projection came out wrong in 4 of 6 runs; the same kernel without spills: 0 of 8.  The product's
scratch users are fi_bwd_taps_c4n and fi_bwd_image_owner (the many-channel backward): stream A runs that backward (C = 8)
in a loop, stream B the projection and the RGB backward; B's results are compared with the ones computed alone."""
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
ta = synth.torch_inputs(dev, 2, 8, 360, 640, flow_kind="smooth", seed=1, with_grad=True)
tb = synth.torch_inputs(dev, 2, 3, 128, 256, flow_kind="smooth", seed=2, with_grad=True)
tc = synth.torch_inputs(dev, 1, 12, 200, 320, flow_kind="smooth", seed=3, with_grad=True)
a1, a2, a3 = torch.zeros_like(ta["x"]), torch.zeros_like(ta["flow"]), torch.zeros_like(ta["filt"])


def work_a():
    assert L.FilterInterpolationLayer_gpu_backward(ta["x"], ta["flow"], ta["filt"], ta["gout"], a1, a2, a3) == 0


def work_b():
    f = tb["flow"]
    cnt, out = f.new_zeros((2, 1, 128, 256)), torch.zeros_like(f)
    assert L.FlowProjectionLayer_gpu_forward(f, cnt, out, 1) == 0
    g1, g2, g3 = torch.zeros_like(tb["x"]), torch.zeros_like(f), torch.zeros_like(tb["filt"])
    assert L.FilterInterpolationLayer_gpu_backward(tb["x"], f, tb["filt"], tb["gout"], g1, g2, g3) == 0
    h1, h2, h3 = torch.zeros_like(tc["x"]), torch.zeros_like(tc["flow"]), torch.zeros_like(tc["filt"])
    assert L.FilterInterpolationLayer_gpu_backward(tc["x"], tc["flow"], tc["filt"], tc["gout"], h1, h2, h3) == 0   # spills, too
    return cnt, out, g1, g2, g3, h1, h2, h3


work_a()
ref_a = [t.clone() for t in (a1, a2, a3)]
ref_b = [t.clone() for t in work_b()]
torch.cuda.synchronize()
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
bad_a = bad_b = 0
for it in range(200):
    torch.cuda.synchronize()
    with torch.cuda.stream(sa):
        for _ in range(3):
            work_a()
    with torch.cuda.stream(sb):
        got = work_b()
    torch.cuda.synchronize()
    if any((g - r).abs().max().item() > 1e-3 * max(1.0, r.abs().max().item()) for g, r in zip(got, ref_b)):
        bad_b += 1
    if any((g - r).abs().max().item() > 1e-3 * max(1.0, r.abs().max().item()) for g, r in zip((a1, a2, a3), ref_a)):
        bad_a += 1
print("200 rounds of (many-channel backward x 3 on stream A) || (projection + RGB backward + many-channel backward on stream B): "
      "B wrong in %d, A wrong in %d" % (bad_b, bad_a))
