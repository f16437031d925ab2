"""Forward warps at other channel counts than the headline (C = 64 context features, odd counts): quick timing
of InterpolationCh and FilterInterpolation through the C ABI.  Usage: python tools/bench_channels.py (needs a GPU)."""
import sys, torch, time
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'memc-net_amd'))
import my_package._ext.my_lib as L
from tools import synth
dev=torch.device('cuda:0')
for (B,C) in ((8,64),(8,5),(32,3)):
    t=synth.torch_inputs(dev,B,C,720,1280,flow_kind='smooth')
    out=torch.empty_like(t['x'])
    fn=(lambda: L.InterpolationChLayer_gpu_forward(t['x'],t['flow'],out))
    for _ in range(30): fn()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): fn()
    e1.record(); torch.cuda.synchronize()
    us=e0.elapsed_time(e1)*1e3/50; by=B*720*1280*(C*8+8)
    print("InterpolationCh fwd B=%d C=%d: %.1f us  %.0f GB/s (%.1f%% of 8 TB/s)"%(B,C,us,by/us/1e3,by/us/1e3/80))
    fn2=(lambda: L.FilterInterpolationLayer_gpu_forward(t['x'],t['flow'],t['filt'],out))
    for _ in range(30): fn2()
    torch.cuda.synchronize(); e0.record()
    for _ in range(50): fn2()
    e1.record(); torch.cuda.synchronize()
    us=e0.elapsed_time(e1)*1e3/50; by=B*720*1280*(C*8+72)
    print("FilterInterpolation fwd B=%d C=%d: %.1f us  %.0f GB/s (%.1f%%)"%(B,C,us,by/us/1e3,by/us/1e3/80))
