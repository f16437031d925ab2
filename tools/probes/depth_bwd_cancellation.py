#!/usr/bin/env python
"""tools/probes/depth_bwd_cancellation.py -- the one random case the 600-case fresh-seed sweep of round 6 tripped on
(3x9x90x130, smooth flow x 40, seed 777002): DepthFlowProjection backward, gradinput2.  A cell is a sum of eight terms
go / count * (f - out) of either sign; at |f| ~ 100 px the terms are ~1e3 and fp32 leaves ~1e-4 of THEM whatever the sum.
Four evaluations of the same call: this library, the reference's own kernel (oracle/_ref), the fp32 oracle, and float64."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for _p in (ROOT, os.path.join(ROOT, "memc-net_amd"), os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)
os.environ.setdefault("MEMC_RANDOM_SEED", "777002")
os.environ.setdefault("MEMC_RANDOM_CASES", "600")
import torch  # noqa: E402

import my_package._ext.my_lib as my_lib  # noqa: E402
from oracle import memc_oracle as O  # noqa: E402
from oracle import ref_gpu as R  # noqa: E402
import test_gpu_parity as TP  # noqa: E402

case = [c for c in TP.RANDOM_CASES if tuple(c[:4]) == (3, 9, 90, 130) and c[4] == "smooth" and c[5] == 40][0]
d = TP._make_random(case)
dev = torch.device("cuda:0")
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)   # noqa: E731
f, dep, gf = T(d["flow"]), T(d["depth"]), T(d["gflow"])
B, _, H, W = f.shape
cnt, out = torch.zeros(B, 1, H, W, device=dev), torch.zeros(B, 2, H, W, device=dev)
assert my_lib.DepthFlowProjectionLayer_gpu_forward(f, dep, cnt, out, 0) == 0
gin, gd = torch.zeros_like(f), torch.zeros_like(dep)
assert my_lib.DepthFlowProjectionLayer_gpu_backward(f, dep, cnt, out, gf, gin, gd) == 0
want_out, want_cnt = O.depth_flow_projection_forward(d["flow"], d["depth"], 0)
w1, w2 = O.depth_flow_projection_backward(d["flow"], d["depth"], want_cnt, want_out, d["gflow"])
hip = gd.cpu().numpy().astype(np.float64)
orc = w2.astype(np.float64)
# float64 evaluation of the reference's formula (my_lib.c:1805-1873) from the SAME forward planes (the oracle's)
fl, de, co, ou, go = (a.astype(np.float64) for a in (d["flow"], d["depth"], want_cnt, want_out, d["gflow"]))
ex = np.zeros_like(de)
mag = np.zeros_like(de)
for b in range(B):
    for y in range(H):
        for x in range(W):
            fx, fy = fl[b, 0, y, x], fl[b, 1, y, x]
            x2, y2 = np.float32(x) + np.float32(fl[b, 0, y, x]), np.float32(y) + np.float32(fl[b, 1, y, x])   # positions in fp32 as the C does
            if x2 < 0 or y2 < 0 or x2 > W - 1 or y2 > H - 1:
                continue
            L, Tt = int(x2), int(y2)
            Rr, Bm = min(L + 1, W - 1), min(Tt + 1, H - 1)
            s = m = 0.0
            for (yy, xx) in ((Tt, L), (Tt, Rr), (Bm, L), (Bm, Rr)):
                for kk, fk in ((0, fx), (1, fy)):
                    t = go[b, kk, yy, xx] / co[b, 0, yy, xx] * (fk - ou[b, kk, yy, xx])
                    s -= t
                    m += abs(t)
            ex[b, 0, y, x] = s
            mag[b, 0, y, x] = m
ref = None
if R.available():
    r1, r2 = R.depth_flow_projection_backward(f, dep, T(want_cnt), T(want_out), gf)
    ref = r2.cpu().numpy().astype(np.float64)
print("case", case[:6], " |exact| up to %.4g, sum of |terms| up to %.4g" % (np.abs(ex).max(), mag.max()))
for name, a in (("this library (HIP)", hip), ("fp32 oracle (CPU port)", orc), ("reference kernel (oracle/_ref)", ref)):
    if a is None:
        continue
    e = np.abs(a - ex)
    i = np.unravel_index(np.argmax(e), e.shape)
    print("%-32s max |err vs float64| %.3g at %s (exact %.6g, sum of |terms| there %.4g; err / terms %.2g)" % (
        name, e.max(), i, ex[i], mag[i], e[i] / max(mag[i], 1e-30)))
e = np.abs(hip - orc); i = np.unravel_index(np.argmax(e), e.shape)
print("HIP vs fp32 oracle: max |diff| %.3g at %s: want %.6g, sum of |terms| %.4g" % (e.max(), i, orc[i], mag[i]))
if ref is not None:
    e = np.abs(ref - orc); i = np.unravel_index(np.argmax(e), e.shape)
    print("reference kernel vs fp32 oracle: max |diff| %.3g at %s: want %.6g, sum of |terms| %.4g" % (e.max(), i, orc[i], mag[i]))
    e = np.abs(hip - ref); print("HIP vs reference kernel: max |diff| %.3g" % e.max())
