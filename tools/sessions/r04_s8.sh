#!/bin/bash
# Round 4, GPU session 8: config 2's third attempt (the halves at three workgroups per CU, spilling); the bilinear warp's RGB backward
# through the owner-computes kernels (bl_cap 5) against the packed planes.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04_s8
mkdir -p "$OUT"
cd "$REPO"
if [ -z "${SKIP_PART3:-}" ]; then
echo "== the halves at three per CU (168 VGPRs, spills): one kernel (60) against the halves (62)"
timeout 600 python tools/ab_variants.py --lib memc-net_amd/lib/libmemc_hip_measure_part3.so --op fi_bwd --variants=60,62 --cases fi_bwd_c2,fi_bwd --flows smooth 2>&1 | grep -v amdgpu.ids | tee "$OUT/ab_fi_bwd_halves_three_per_cu.txt"
fi
echo "== bilinear RGB backward: packed planes (-1) against the owner-computes kernels of the many-channel path at C = 3 (5)"
timeout 600 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee "$OUT/ab_bl_bwd_owner.txt"
import sys, os, statistics
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "memc-net_amd"))
import numpy as np, torch
from tools import measure as M, synth
M.use(); L = M.bound()
from oracle import memc_oracle as O
dev = torch.device("cuda:0")
# parity of the arm first (small case)
rng = np.random.default_rng(3)
xn, fn, gn = synth.np_image(rng, 2, 3, 80, 192), synth.np_flow(rng, 2, 80, 192, "smooth", 5.0), synth.np_image(rng, 2, 3, 80, 192)
w1, w2 = O.interpolation_backward(xn, fn, gn)
for cap in (-1, 5):
    M.set_variant("bl_cap", cap)
    h1, h2 = torch.zeros(xn.shape, device=dev), torch.zeros(fn.shape, device=dev)
    rc = L.InterpolationLayer_gpu_backward(torch.from_numpy(xn).to(dev), torch.from_numpy(fn).to(dev), torch.from_numpy(gn).to(dev), h1, h2)
    import ctypes
    lp = M.lib().memc_debug_last_path; lp.restype = ctypes.c_char_p
    print("bl_cap %2d: rc %d path %s  max err gradinput1 %.3g gradinput2 %.3g" % (cap, rc, lp().decode(),
          float(np.abs(h1.cpu().numpy() - w1).max()), float(np.abs(h2.cpu().numpy() - w2).max())))
for flow in ("smooth", "iid"):
    t = synth.torch_inputs(dev, 32, 3, 720, 1280, flow_kind=flow, with_grad=True)
    x, f, g = t["x"], t["flow"], t["gout"]
    g1, g2 = torch.zeros_like(x), torch.zeros_like(f)
    ts = {-1: [], 5: []}
    for r in range(5):
        for cap in (-1, 5):
            M.set_variant("bl_cap", cap)
            for _ in range(3): L.InterpolationLayer_gpu_backward(x, f, g, g1, g2)
            for _ in range(10):
                g1.zero_()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); L.InterpolationLayer_gpu_backward(x, f, g, g1, g2); e1.record(); e1.synchronize()
                ts[cap].append(e0.elapsed_time(e1) * 1e3)
    M.set_variant("bl_cap", -1)
    print("bl_bwd flow=%-6s packed planes %8.1f us   owner kernels at C = 3 %8.1f us" % (flow, statistics.median(ts[-1]), statistics.median(ts[5])))
PY
