"""How does the bilinear forward (84 VGPRs; 48 KiB of LDS -> 3 workgroups per CU) respond to FEWER workgroups per CU?
Pads its LDS request: +8 KiB -> 2 per CU, +40 KiB -> 1 per CU.  If time scales with 1 / occupancy the kernel is
latency-bound and a smaller staging budget (more workgroups per CU) is the lever."""
import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "memc-net_amd"))
import my_package._ext.my_lib as L
from tools import measure as M  # noqa: E402
M.use()                             # the measurement build: ablation / A-B arms live only there
from tools import synth
dev = torch.device("cuda:0")
t = synth.torch_inputs(dev, 32, 3, 720, 1280, flow_kind="smooth")
x, f = t["x"], t["flow"]
o = torch.empty_like(x)
fn = lambda: L.InterpolationLayer_gpu_forward(x, f, o)
for _ in range(200): fn()
for extra, per_cu in ((0, 3), (8 << 10, 2), (40 << 10, 1), (0, 3)):
    M.set_variant("extra_lds", extra)
    for _ in range(30): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(50): fn()
    e1.record(); torch.cuda.synchronize()
    print("bl_fwd_tiled<3>  %d workgroup(s) per CU: %.1f us" % (per_cu, e0.elapsed_time(e1) * 1e3 / 50))
M.set_variant("extra_lds", 0)
