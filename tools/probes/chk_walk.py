"""A/B of the tile walk (strips vs stripes of n tile columns per XCD) for the kernels that take it at run time:
results must be identical (forward, assigned gradients) or equal up to atomic order; then timings at 720p."""
import sys, os, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "memc-net_amd"))
import my_package._ext.my_lib as L
from tools import measure as M  # noqa: E402
M.use()                             # the measurement build: ablation / A-B arms live only there
from tools import synth
dev = torch.device("cuda:0")


def run(t, cnt, pout):
    x, f, g, d = t["x"], t["flow"], t["gout"], t["depth"]
    gf = t["gflow"]
    o = torch.full_like(x, float("nan")); L.InterpolationLayer_gpu_forward(x, f, o)
    g1, g2 = torch.zeros_like(x), torch.full_like(f, float("nan")); L.InterpolationLayer_gpu_backward(x, f, g, g1, g2)
    p1 = torch.full_like(f, float("nan")); L.FlowProjectionLayer_gpu_backward(f, cnt, gf, p1)
    q1, q2 = torch.full_like(f, float("nan")), torch.full_like(d, float("nan"))
    L.DepthFlowProjectionLayer_gpu_backward(f, d, cnt, pout, gf, q1, q2)
    return o, g1, g2, p1, q1, q2


for (B, H, W) in ((2, 720, 1280), (3, 50, 200), (1, 33, 456), (5, 16, 64)):
    t = synth.torch_inputs(dev, B, 3, H, W, flow_kind="smooth", with_grad=True, with_depth=True)
    t["gflow"] = torch.rand_like(t["flow"])
    cnt, pout = torch.empty_like(t["depth"]), torch.empty_like(t["flow"])
    L.DepthFlowProjectionLayer_gpu_forward(t["flow"], t["depth"], cnt, pout, 0)
    M.set_variant("walk", 0); ref = run(t, cnt, pout)
    for sw in (2, 4, 5):
        M.set_variant("walk", sw); got = run(t, cnt, pout)
        errs = [float((a - b).abs().max()) for a, b in zip(ref, got)]
        print((B, H, W), "sw", sw, "max diffs", ["%.1e" % e for e in errs], "nan", any(bool(torch.isnan(a).any()) for a in got))
M.set_variant("walk", -1)

t = synth.torch_inputs(dev, 32, 3, 720, 1280, flow_kind="smooth", with_grad=True, with_depth=True)
t["gflow"] = torch.rand_like(t["flow"])
cnt, pout = torch.empty_like(t["depth"]), torch.empty_like(t["flow"])
L.DepthFlowProjectionLayer_gpu_forward(t["flow"], t["depth"], cnt, pout, 0)
x, f, g, d, gf = t["x"], t["flow"], t["gout"], t["depth"], t["gflow"]
o, g1, g2 = torch.empty_like(x), torch.zeros_like(x), torch.empty_like(f)
p1, q2 = torch.empty_like(f), torch.empty_like(d)
ops = {"interp_fwd": lambda: L.InterpolationLayer_gpu_forward(x, f, o),
       "interp_bwd": lambda: L.InterpolationLayer_gpu_backward(x, f, g, g1, g2),
       "proj_bwd": lambda: L.FlowProjectionLayer_gpu_backward(f, cnt, gf, p1),
       "dproj_bwd": lambda: L.DepthFlowProjectionLayer_gpu_backward(f, d, cnt, pout, gf, p1, q2)}
for rep in range(2):
    for sw in (0, 2, 4, 0, 2, 4):
        M.set_variant("walk", sw)
        line = []
        for name, fn in ops.items():
            for _ in range(40): fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for _ in range(40): fn()
            e1.record(); torch.cuda.synchronize()
            line.append("%s %.1f" % (name, e0.elapsed_time(e1) * 1e3 / 40))
        print("sw=%d  " % sw + "  ".join(line))
    break
M.set_variant("walk", -1)
