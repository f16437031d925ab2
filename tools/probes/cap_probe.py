"""2x2-footprint kernels (bilinear fwd / bwd, projection bwd) with a 39 KiB instead of 48 KiB LDS staging budget
(4 instead of 3 workgroups per CU): differences vs the 48 KiB results, and timings at 720p batch 32."""
import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "memc-net_amd"))
import my_package._ext.my_lib as L
from tools import measure as M  # noqa: E402
M.use()                             # the measurement build: ablation / A-B arms live only there
from tools import synth
dev = torch.device("cuda:0")
for kind in ("smooth", "video"):
    t = synth.torch_inputs(dev, 32, 3, 720, 1280, flow_kind=kind, with_grad=True, with_depth=True)
    x, f, g, d = t["x"], t["flow"], t["gout"], t["depth"]
    gf = torch.rand_like(f)
    cnt, pout = torch.empty_like(d), torch.empty_like(f)
    L.DepthFlowProjectionLayer_gpu_forward(f, d, cnt, pout, 0)
    res = {}
    for cap in (0, 1, 2, 1, 2):
        M.set_variant("bl_cap", cap)
        o, g1, g2 = torch.empty_like(x), torch.zeros_like(x), torch.empty_like(f)
        p1, q1, q2 = torch.empty_like(f), torch.empty_like(f), torch.empty_like(d)
        ops = {"interp_fwd": lambda: L.InterpolationLayer_gpu_forward(x, f, o),
               "interp_bwd": lambda: L.InterpolationLayer_gpu_backward(x, f, g, g1, g2),
               "proj_bwd": lambda: L.FlowProjectionLayer_gpu_backward(f, cnt, gf, p1),
               "dproj_bwd": lambda: L.DepthFlowProjectionLayer_gpu_backward(f, d, cnt, pout, gf, q1, q2)}
        line = []
        for name, fn in ops.items():
            for _ in range(40): fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for _ in range(40): fn()
            e1.record(); torch.cuda.synchronize()
            line.append("%s %.1f" % (name, e0.elapsed_time(e1) * 1e3 / 40))
        g1.zero_(); ops["interp_bwd"](); torch.cuda.synchronize()
        cur = [o, g1, g2, p1, q1, q2]
        if cap not in res:
            res[cap] = [a.clone() for a in cur]
        diffs = ["%.1e" % float((a - b).abs().max()) for a, b in zip(cur, res[0])]
        print("flow=%-6s budget %s  %s   max diff vs 48 KiB %s" % (kind, ("48K", "39K", "31K")[cap], "  ".join(line), diffs))
M.set_variant("bl_cap", -1)
