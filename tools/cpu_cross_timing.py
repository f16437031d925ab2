#!/usr/bin/env python
"""tools/cpu_cross_timing.py -- BUILD CONTAINER ONLY, TIMING ONLY: is the oracle -- the port
that bench.py times as `cpu_baseline` (kind "port") -- a fair stand-in for the reference's own CPU code?  Times the
reference's my_package/src/my_lib.c, compiled from where it lies, next to oracle/memc_oracle.c, both single-threaded (the
reference's loops are serial as written), on the same arrays, and prints Mpixel/s for the FilterInterpolation forward
and backward and the FlowProjection forward.

This is NOT a parity pin and nothing else in the repository uses it: my_lib.c includes <TH.h> of PyTorch 0.2, which the
image lacks, so the translation unit is given the three declarations it uses (a struct with size / stride / data and
THFloatTensor_data) on the command line of THIS script, in a temporary directory -- enough to time its loops, and by the
rule of oracle/ not an admissible reference build (the oracle is pinned against the reference's GPU kernels instead,
DESIGN.md section 2).  Nothing is written into the repository except the numbers (profiles/r04_cpu_cross_timing.txt).

    python tools/cpu_cross_timing.py [--out profiles/r04_cpu_cross_timing.txt]
"""
import argparse
import ctypes
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF_C = "/root/reference/my_package/src/my_lib.c"

TH_DECLS = """
#ifndef MEMC_TIMING_TH_H
#define MEMC_TIMING_TH_H
typedef struct THFloatTensor { long *size; long *stride; int nDimension; float *data; } THFloatTensor;
static inline float *THFloatTensor_data(THFloatTensor *t) { return t->data; }
#endif
"""


class TH(ctypes.Structure):
    _fields_ = [("size", ctypes.POINTER(ctypes.c_long)), ("stride", ctypes.POINTER(ctypes.c_long)),
                ("nDimension", ctypes.c_int), ("data", ctypes.POINTER(ctypes.c_float))]


def th(a):
    t = TH()
    t._keep = (a, (ctypes.c_long * 4)(*a.shape), (ctypes.c_long * 4)(*[s // 4 for s in a.strides]))
    t.size, t.stride, t.nDimension = t._keep[1], t._keep[2], 4
    t.data = a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))
    return t


def best(fn, reps):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return min(ts)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r04_cpu_cross_timing.txt"))
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    if not os.path.exists(REF_C):
        raise SystemExit("needs /root/reference (build container only)")
    os.environ["OMP_NUM_THREADS"] = "1"                      # before the oracle's OpenMP runtime starts
    from oracle import memc_oracle as O
    O.build()
    tmp = tempfile.mkdtemp(prefix="memc_cross_timing_")
    open(os.path.join(tmp, "TH.h"), "w").write(TH_DECLS)
    so = os.path.join(tmp, "libref_c_timing.so")
    subprocess.run(["gcc", "-O2", "-shared", "-fPIC", "-w", "-I", tmp, "-o", so, REF_C, "-lm"], check=True)
    ref = ctypes.CDLL(so)
    B, C, H, W = 1, 3, 720, 1280
    rng = np.random.default_rng(0)
    from tools import synth
    x, flow, filt, gout = (np.ascontiguousarray(v) for v in (synth.np_image(rng, B, C, H, W), synth.np_flow(rng, B, H, W, "smooth"),
                                                           synth.np_filter(rng, B, H, W), synth.np_image(rng, B, C, H, W)))
    sites = B * H * W
    lines = ["CPU cross-timing, single thread, %dx%dx%dx%d, smooth flow (build container: %d cores, %s); best of %d" % (
        B, C, H, W, os.cpu_count(), subprocess.run(["gcc", "--version"], capture_output=True, text=True).stdout.split("\n")[0], a.reps),
        "reference = /root/reference/my_package/src/my_lib.c (gcc -O2; timing only, see the script's header); port = oracle/memc_oracle.c "
        "(OMP_NUM_THREADS=1)", "%-34s %14s %14s %8s" % ("operator", "reference Mpix/s", "port Mpix/s", "port/ref")]

    def row(name, t_ref, t_port):
        lines.append("%-34s %14.2f %14.2f %8.2f" % (name, sites / t_ref / 1e6, sites / t_port / 1e6, t_ref / t_port))

    out = np.zeros_like(x)
    assert ref.FilterInterpolationLayer_cpu_forward(ctypes.byref(th(x)), ctypes.byref(th(flow)), ctypes.byref(th(filt)), ctypes.byref(th(out))) == 0
    t_ref = best(lambda: ref.FilterInterpolationLayer_cpu_forward(ctypes.byref(th(x)), ctypes.byref(th(flow)), ctypes.byref(th(filt)), ctypes.byref(th(out))), a.reps)
    t_port = best(lambda: O.filter_interpolation_forward(x, flow, filt), a.reps)
    same = float(np.abs(out - O.filter_interpolation_forward(x, flow, filt)).max())
    row("FilterInterpolation forward", t_ref, t_port)
    g1, g2, g3 = np.zeros_like(x), np.zeros_like(flow), np.zeros_like(filt)

    def ref_bwd():
        g1[...] = 0; g2[...] = 0; g3[...] = 0
        ref.FilterInterpolationLayer_cpu_backward(ctypes.byref(th(x)), ctypes.byref(th(flow)), ctypes.byref(th(filt)), ctypes.byref(th(gout)),
                                                   ctypes.byref(th(g1)), ctypes.byref(th(g2)), ctypes.byref(th(g3)))
    row("FilterInterpolation backward", best(ref_bwd, a.reps), best(lambda: O.filter_interpolation_backward(x, flow, filt, gout), a.reps))
    cnt, po = np.zeros((B, 1, H, W), np.float32), np.zeros_like(flow)

    def ref_proj():
        cnt[...] = 0; po[...] = 0
        ref.FlowProjectionLayer_cpu_forward(ctypes.byref(th(flow)), ctypes.byref(th(cnt)), ctypes.byref(th(po)), 0)
    row("FlowProjection forward (fill 0)", best(ref_proj, a.reps), best(lambda: O.flow_projection_forward(flow, 0), a.reps))
    lines.append("(the two agree on this input to %.3g -- informational: the oracle's pin is the reference's GPU kernels)" % same)
    text = "\n".join(lines) + "\n"
    print(text, end="")
    open(a.out, "w").write(text)


if __name__ == "__main__":
    main()
