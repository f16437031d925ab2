import os, sys
ROOT = os.getcwd()
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "memc-net_amd"))
import torch
from tools import synth
from tools.ab_libs import Bound
a, b = Bound(os.path.abspath(sys.argv[1])), Bound(os.path.abspath(sys.argv[2]))
dev = torch.device("cuda:0")
t = synth.torch_inputs(dev, 32, 3, 720, 1280, flow_kind="smooth", with_depth=True)
for pan in (0.0, 40.0):
    f = t["flow"].clone(); f[:, 0] += pan; f[:, 1] -= pan / 2
    res = []
    for l in (a, b):
        cnt, out = f.new_zeros((32, 1, 720, 1280)), torch.zeros_like(f)
        l.FlowProjectionLayer_gpu_forward(f, cnt, out, 1)
        res.append((cnt.clone(), out.clone()))
    print("pan %g: count equal %s, max |out diff| %.3g" % (pan, torch.equal(res[0][0], res[1][0]), float((res[0][1] - res[1][1]).abs().max())))
