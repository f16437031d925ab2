#!/usr/bin/env python
"""Race screen: the tiled projection forward vs the scalar reference-shaped kernels on the same GPU, many
random inputs; count must match bit-for-bit, out within 1e-4."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "memc-net_amd")):
    sys.path.insert(0, p)
import torch
import my_package._ext.my_lib as L
from tools import measure as M  # noqa: E402
M.use()                             # the measurement build: ablation / A-B arms live only there

dev = torch.device("cuda:0")
# automatic, general path, owner geometries (tile height 16 / 32 / 64, strips / stripes), the round-1 owner kernel
OTHERS = (-1, 1, 100, 104, 110, 112, -40, 402, 414, 130, 144, 150, 164, -10)
bad = 0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 60):
    g = torch.Generator(device=dev); g.manual_seed(it)
    B, H, W = 1 + it % 3, 16 * (1 + it % 9) + (it % 5), 64 * (1 + it % 4) + 4 * (it % 7)
    sig = [1.0, 3.0, 8.0, 20.0][it % 4]
    f = torch.randn(B, 2, H, W, device=dev, generator=g) * sig
    others = OTHERS
    # Round 4: structured LARGE motion every second iteration (the far path: landing boxes, stamps, tiles without near
    # sources) on images of many tiles -- a camera pan, a fast rectangle over a slow background, a few far sources, motion
    # right at the reach of 24 px -- the product's kernels and two other geometries against the scalar kernels
    kind = ("iid", "pan", "object", "sparse", "threshold")[(it // 2) % 5] if it % 2 else "iid"
    if kind != "iid":
        H, W = 40 + 37 * (it % 6), 200 + 92 * (it % 5)
        f = torch.randn(B, 2, H, W, device=dev, generator=g) * [0.3, 2.0, 5.0][it % 3]
        r = lambda lo, hi: float(torch.rand((), device=dev, generator=g)) * (hi - lo) + lo      # noqa: E731
        if kind == "pan":
            f[:, 0] += r(-90, 90)
            f[:, 1] += r(-60, 60)
        elif kind == "object":
            y0, x0 = int(r(0, H - 30)), int(r(0, W - 90))
            f[:, 0, y0:y0 + 30 + it % 40, x0:x0 + 90 + it % 70] = r(-120, 120)
            f[:, 1, y0:y0 + 30 + it % 40, x0:x0 + 90 + it % 70] = r(-80, 80)
        elif kind == "sparse":
            for _ in range(1 + it % 7):
                f[int(r(0, B)), :, int(r(0, H)), int(r(0, W))] = torch.tensor([r(-W, W), r(-H, H)], device=dev)
        else:
            pick = torch.randint(0, 6, (B, 2, H, W), device=dev, generator=g)
            vals = torch.tensor([23.99, 24.0, -24.0, 24.01, -23.5, 0.5], device=dev)
            f = torch.where(torch.rand((B, 2, H, W), device=dev, generator=g) < 0.02, vals[pick], f)
        others = (-1, 1, 104, 112)
    d = torch.rand(B, 1, H, W, device=dev, generator=g) + 0.1
    res = {}
    for v in (0,) + others:
        M.set_variant("projection", v)
        for fh in (0, 1):
            c = f.new_zeros(B, 1, H, W); o = torch.zeros_like(f)
            assert L.FlowProjectionLayer_gpu_forward(f, c, o, fh) == 0
            c2 = f.new_zeros(B, 1, H, W); o2 = torch.zeros_like(f)
            assert L.DepthFlowProjectionLayer_gpu_forward(f, d, c2, o2, fh) == 0
            res[(v, fh)] = (c, o, c2, o2)
    torch.cuda.synchronize()
    for fh, other in [(fh, o) for o in others for fh in (0, 1)]:
        a, b = res[(0, fh)], res[(other, fh)]
        ok = torch.equal(a[0], b[0]) and (a[1] - b[1]).abs().max().item() <= 1e-4 and \
            (a[2] - b[2]).abs().max().item() <= 1e-4 and (a[3] - b[3]).abs().max().item() <= 2e-4
        if not ok:
            bad += 1
            print("MISMATCH variant=%d it=%d kind=%s B=%d H=%d W=%d sig=%g fh=%d: count eq %s, out err %.3g, dcount err %.3g, dout err %.3g" % (
                other, it, kind, B, H, W, sig, fh, torch.equal(a[0], b[0]), (a[1] - b[1]).abs().max().item(),
                (a[2] - b[2]).abs().max().item(), (a[3] - b[3]).abs().max().item()))
            if not torch.equal(a[0], b[0]):
                idx = (a[0] != b[0]).nonzero()
                print("   count diffs:", idx.shape[0], idx[:5].tolist(), a[0][a[0] != b[0]][:5].tolist(), b[0][a[0] != b[0]][:5].tolist())
M.set_variant("projection", -1)
print("stress done, mismatches:", bad)
