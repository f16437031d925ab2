"""FETCH_SIZE of the bilinear forward under the strip walk vs stripes (3 tile columns: a different grid size, so the
two groups separate in the counter table).  Run under:  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d D -o r -- python this
then:  python this --report D/r_results.db"""
import sys, os
if len(sys.argv) > 2 and sys.argv[1] == "--report":
    import sqlite3
    cur = sqlite3.connect(sys.argv[2]).cursor()
    for row in cur.execute("select kernel_name, grid_size, count(*), avg(value), avg(duration) from counters_collection "
                           "where kernel_name like '%tiled%' group by kernel_name, grid_size"):
        print("%-60s grid %9d  n=%3d  FETCH_SIZE %.0f KiB (x2 = %.0f MB)  %.1f us" % (
            row[0].split("(")[0][-60:], row[1], row[2], row[3], row[3] * 2 * 1024 / 1e6, row[4] / 1e3))
    sys.exit(0)
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "memc-net_amd"))
import my_package._ext.my_lib as L
from tools import measure as M  # noqa: E402
M.use()                             # the measurement build: ablation / A-B arms live only there
from tools import synth
dev = torch.device("cuda:0")
t = synth.torch_inputs(dev, 32, 3, 720, 1280, flow_kind="smooth", with_grad=True, with_depth=True)
x, f, g, d = t["x"], t["flow"], t["gout"], t["depth"]
gf = torch.rand_like(f)
cnt, pout = torch.empty_like(d), torch.empty_like(f)
L.DepthFlowProjectionLayer_gpu_forward(f, d, cnt, pout, 0)
o, g1, g2, p1 = torch.empty_like(x), torch.zeros_like(x), torch.empty_like(f), torch.empty_like(f)
for sw in (0, 3):
    M.set_variant("walk", sw)
    for _ in range(10):
        L.InterpolationLayer_gpu_forward(x, f, o)
        L.InterpolationLayer_gpu_backward(x, f, g, g1, g2)
        L.FlowProjectionLayer_gpu_backward(f, cnt, gf, p1)
torch.cuda.synchronize()
