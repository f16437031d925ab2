#!/usr/bin/env python
"""tools/probes/proj_motion_sweep.py -- FlowProjection forward (with hole filling) against the size of the motion: the
benchmark's smooth flow scaled by 1, 1.5, 2, 3, 4, 6 (|flow| up to ~18 px at scale 1).  Sources that move 24 px or more
are "far": their images are redone by proj_owner_far -- what does that cost, and from which motion on?"""
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
t = synth.torch_inputs(dev, 32, 3, 720, 1280, flow_kind="smooth", with_depth=True)
f0, d = t["flow"], t["depth"]
cnt, out = f0.new_zeros((32, 1, 720, 1280)), torch.zeros_like(f0)
for _ in range(100):
    L.FlowProjectionLayer_gpu_forward(f0, cnt, out, 0)
print("%-7s %10s %12s %14s %14s %16s" % ("scale", "max |f|", "far sites %", "fill 0, us", "fill 1, us", "depth fill 1, us"))
for scale in (1.0, 1.5, 2.0, 3.0, 4.0, 6.0):
    f = (f0 * scale).contiguous()
    far = float(((f.abs() >= 24).any(dim=1)).float().mean()) * 100
    row = []
    for fn in (lambda: L.FlowProjectionLayer_gpu_forward(f, cnt, out, 0), lambda: L.FlowProjectionLayer_gpu_forward(f, cnt, out, 1),
               lambda: L.DepthFlowProjectionLayer_gpu_forward(f, d, cnt, out, 1)):
        for _ in range(3):
            fn()
        ts = []
        for _ in range(8):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); e1.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts.sort()
        row.append(ts[len(ts) // 2])
    print("%-7.1f %10.1f %12.3f %14.1f %14.1f %16.1f" % (scale, float(f.abs().max()), far, row[0], row[1], row[2]))

print()
print("camera pans (p, -p/2) on top of the benchmark's flow (round 5: the scan is shifted by the image's dominant motion)")
print("%-7s %14s %14s %16s" % ("pan px", "fill 0, us", "fill 1, us", "depth fill 1, us"))
for pan in (0.0, 3.0, 8.0, 20.0, 40.0, 80.0, 160.0):
    f = f0.clone()
    f[:, 0] += pan
    f[:, 1] -= pan / 2
    row = []
    for fn in (lambda: L.FlowProjectionLayer_gpu_forward(f, cnt, out, 0), lambda: L.FlowProjectionLayer_gpu_forward(f, cnt, out, 1),
               lambda: L.DepthFlowProjectionLayer_gpu_forward(f, d, cnt, out, 1)):
        for _ in range(3):
            fn()
        ts = []
        for _ in range(8):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); e1.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts.sort()
        row.append(ts[len(ts) // 2])
    print("%-7.1f %14.1f %14.1f %16.1f" % (pan, row[0], row[1], row[2]))
sys.stdout.flush()


def far_model(f, images=4, TH=32, R=24):
    """What proj_owner_far decides for the first `images` images of f (numpy restatement of its culling): the share of tiles
    it recomputes, and per recomputed tile the source quads its waves scan, in units of the owner kernel's scan
    (128 x 81 sources)."""
    import numpy as np
    fn = f[:images].cpu().numpy()
    B, _, H, W = fn.shape
    ntx, nty = (W + 63) // 64, (H + TH - 1) // TH
    redo = scanned = 0
    for b in range(B):
        fx, fy = fn[b, 0], fn[b, 1]
        ys, xs = np.mgrid[0:H, 0:W].astype(np.float32)
        x2, y2 = xs + fx, ys + fy
        valid = ~((np.abs(fx) < R) & (np.abs(fy) < R)) & (x2 >= 0) & (y2 >= 0) & (x2 <= W - 1) & (y2 <= H - 1)
        box = np.full((nty, ntx, 4), np.nan, np.float32)
        for ty in range(nty):
            for tx in range(ntx):
                v = valid[ty * TH:(ty + 1) * TH, tx * 64:(tx + 1) * 64]
                if v.any():
                    a, c = x2[ty * TH:(ty + 1) * TH, tx * 64:(tx + 1) * 64][v], y2[ty * TH:(ty + 1) * TH, tx * 64:(tx + 1) * 64][v]
                    box[ty, tx] = (a.min(), a.max(), c.min(), c.max())
        for ty in range(nty):
            for tx in range(ntx):
                tx0, ty0 = tx * 64, ty * TH
                hit = (box[..., 1] >= tx0 - 1) & (box[..., 0] < tx0 + 64) & (box[..., 3] >= ty0 - 1) & (box[..., 2] < ty0 + TH)
                other = hit.copy(); other[ty, tx] = False
                if not other.any():
                    continue
                redo += 1
                rows = 0
                for sty in range(nty):
                    for stx in range(ntx):
                        sx0 = stx * 64
                        if hit[sty, stx]:
                            rows += TH
                            continue
                        if not (sx0 + 64 + R >= tx0 - 1 and sx0 - R - 1 < tx0 + 64):
                            continue
                        for w0 in range(0, TH, 4):
                            sy0 = sty * TH + w0
                            rows += 4 * (sy0 + 4 + R >= ty0 - 1 and sy0 - R - 1 < ty0 + TH)
                scanned += rows * 64
    return 100.0 * redo / (B * ntx * nty), scanned / max(redo, 1) / (128 * 81)


print()
print("the far kernel's decisions (first 4 images): tiles recomputed, sources scanned per recomputed tile / the owner kernel's 128 x 81")
for scale in (1.5, 2.0, 3.0, 4.0, 6.0):
    print("%-7.1f %8.1f %% %8.2f x" % ((scale,) + far_model(f0 * scale)))
