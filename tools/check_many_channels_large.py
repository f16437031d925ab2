"""tools/check_many_channels_large.py -- the many-channel backward passes (fi_bwd_cn.hip) at 4K / 1080p sizes against the
reference's own kernels (oracle/_ref): FilterInterpolation and InterpolationCh, C = 8 / 12 / 6, NaN-filled gradient buffers.
Run on the GPU box; prints the maximum errors."""
import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'memc-net_amd')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import my_package._ext.my_lib as L
from oracle import ref_gpu as R
from tools import synth
dev = torch.device('cuda:0')
for (B, C, H, W, kind) in ((1, 8, 2160, 3840, 'smooth'), (2, 12, 1080, 1920, 'iid'), (1, 6, 2160, 3840, 'smooth')):
    t = synth.torch_inputs(dev, B, C, H, W, flow_kind=kind, seed=5, with_grad=True)
    x, f, k, g = t['x'], t['flow'], t['filt'], t['gout']
    g1, g2, g3 = (torch.full_like(v, float('nan')) for v in (x, f, k))
    assert L.FilterInterpolationLayer_gpu_backward(x, f, k, g, g1, g2, g3) == 0
    w1, w2, w3 = R.filter_interpolation_backward(x, f, k, g)
    for a, b, n in ((g1, w1, 'g1'), (g2, w2, 'g2'), (g3, w3, 'g3')):
        err = (a.double() - b.double()).abs(); bound = 1e-4 + 5e-5 * b.double().abs()
        print(B, C, H, W, kind, n, 'max err %.3g' % float(err.max()), 'ok' if float((err - bound).max()) <= 0 else 'FAIL')
    h1, h2 = torch.full_like(x, float('nan')), torch.full_like(f, float('nan'))
    assert L.InterpolationChLayer_gpu_backward(x, f, g, h1, h2) == 0
    v1, v2 = R.interpolation_backward(x, f, g, ch=True)
    for a, b, n in ((h1, v1, 'bl g1'), (h2, v2, 'bl g2')):
        err = (a.double() - b.double()).abs(); bound = 1e-4 + 5e-5 * b.double().abs()
        print(B, C, H, W, kind, n, 'max err %.3g' % float(err.max()), 'ok' if float((err - bound).max()) <= 0 else 'FAIL')
