"""Bilinear forward with a smaller LDS staging budget (more workgroups per CU): identical results? faster?"""
import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "memc-net_amd"))
import my_package._ext.my_lib as L
from tools import measure as M  # noqa: E402
M.use()                             # the measurement build: ablation / A-B arms live only there
from tools import synth
dev = torch.device("cuda:0")
for kind in ("smooth", "iid", "video"):
    t = synth.torch_inputs(dev, 32, 3, 720, 1280, flow_kind=kind)
    x, f = t["x"], t["flow"]
    outs = []
    for cap in (0, 1, 2, 0, 1, 2):
        M.set_variant("bl_cap", cap)
        o = torch.full_like(x, float("nan"))
        fn = lambda: L.InterpolationLayer_gpu_forward(x, f, o)
        for _ in range(60): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(50): fn()
        e1.record(); torch.cuda.synchronize()
        outs.append(o)
        print("flow=%-6s budget %s: %.1f us   equal to 48 KiB result: %s" % (
            kind, ("48 KiB (3/CU)", "39 KiB (4/CU)", "31 KiB (5/CU)")[cap], e0.elapsed_time(e1) * 1e3 / 50,
            torch.equal(o, outs[0])))
M.set_variant("bl_cap", -1)
