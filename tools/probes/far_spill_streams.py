#!/usr/bin/env python
"""tools/probes/far_spill_streams.py -- round 4's unexplained failure, re-created and dissected.

Round 4: a build of proj_owner_far at three workgroups per CU (80 VGPRs, 76 B of private scratch per lane) failed
tests/test_gpu_parity.py::test_concurrent_streams_projection in 4 of 6 runs -- reported as "the stream WITHOUT far sources came
out wrong".  That kernel is kept as a measurement arm (projection variant -44; -45: the same kernel on the product's grid).
This script runs that test's two-stream loop under one variable at a time and, for every wrong result, says WHICH stream, which
plane, which cells and what the wrong values look like.

    python tools/probes/far_spill_streams.py [--rounds 12] [--out gpurun_out/far_spill_streams.txt]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for _p in (ROOT, os.path.join(ROOT, "memc-net_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import my_package._ext.my_lib as PL  # noqa: E402  (product library)
from oracle import memc_oracle as O  # noqa: E402  (checker)
from tools import measure as M  # noqa: E402
from tools import synth  # noqa: E402

dev = torch.device("cuda:0")
ML = M.bound()
LAST_NOTES = []
ATOL = 1e-4


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def describe(got_out, got_cnt, want_out, want_cnt, H, W):
    """What a wrong result looks like."""
    d_out = np.abs(got_out - want_out)
    d_cnt = np.abs(got_cnt - want_cnt)
    bad_out = d_out > ATOL * np.maximum(1.0, np.abs(want_out))
    bad_cnt = d_cnt > 0
    info = {"out_cells": int(bad_out.any(axis=1).sum()), "count_cells": int(bad_cnt.sum())}
    cells = np.argwhere(bad_out.any(axis=1) | bad_cnt[:, 0])
    if len(cells):
        b, y, x = cells[:, 0], cells[:, 1], cells[:, 2]
        info["images"] = sorted(set(int(v) for v in b))
        info["box_y"] = [int(y.min()), int(y.max())]
        info["box_x"] = [int(x.min()), int(x.max())]
        info["tiles(b,ty,tx)"] = sorted(set((int(bb), int(yy) // 32, int(xx) // 64) for bb, yy, xx in cells))[:16]
        at_holes = want_cnt[b, 0, y, x] <= 0
        info["of_them_at_holes"] = int(at_holes.sum())
        info["got_zero_there"] = int((np.abs(got_out[b, :, y, x]).max(axis=1) == 0).sum())
        info["got_nan_there"] = int(np.isnan(got_out[b, :, y, x]).any(axis=1).sum())
        def sample(idx):
            return [{"b,y,x": [int(b[i]), int(y[i]), int(x[i])],
                     "got": [float(v) for v in got_out[b[i], :, y[i], x[i]]] + [float(got_cnt[b[i], 0, y[i], x[i]])],
                     "want": [float(v) for v in want_out[b[i], :, y[i], x[i]]] + [float(want_cnt[b[i], 0, y[i], x[i]])]}
                    for i in idx]
        info["first_at_holes"] = sample(np.nonzero(at_holes)[0][:3])
        info["first_not_at_holes"] = sample(np.nonzero(~at_holes)[0][:5])
        info["rows_of_wrong_non_hole_cells"] = sorted(set(int(v) for v in y[~at_holes]))[:12]
    return info


def last_block():
    """(pointer, bytes) of the scratch block the calling thread's last projection call used -- instrumented variants only."""
    import ctypes
    f = getattr(PL._lib, "memc_debug_last_scratch", None) if hasattr(PL, "_lib") else None
    try:
        f = PL._lib.memc_debug_last_scratch
    except AttributeError:
        return None
    p, n = ctypes.c_void_p(), ctypes.c_size_t()
    f.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_size_t)]
    f.restype = None
    f(ctypes.byref(p), ctypes.byref(n))
    out = [p.value, n.value]
    try:
        g = PL._lib.memc_debug_last_scratch2
        a, b, c = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
        g.argtypes = [ctypes.POINTER(ctypes.c_void_p)] * 3
        g.restype = None
        g(ctypes.byref(a), ctypes.byref(b), ctypes.byref(c))
        out += [{"stream_seen_by_library": a.value, "cache_entry": b.value, "entry_stream": c.value}]
    except AttributeError:
        pass
    return out


def one_run(lib_near, lib_far, tn, tf, want_n, want_f, fill, iters, serial, log, streams=None, run=0):
    """tests/test_gpu_parity.py::test_concurrent_streams_projection, instrumented.  Returns (near wrong, far wrong) iterations."""
    s1, s2 = streams if streams else (torch.cuda.Stream(), torch.cuda.Stream())
    H, W = tn.shape[2], tn.shape[3]
    outs, notes = [], []
    for it in range(iters):
        cn, on = tn.new_zeros(tn.shape[0], 1, H, W), torch.zeros_like(tn)
        cf, of = tf.new_zeros(tf.shape[0], 1, H, W), torch.zeros_like(tf)
        torch.cuda.synchronize()
        e0, e1, e2, e3 = (torch.cuda.Event(enable_timing=True) for _ in range(4))
        with torch.cuda.stream(s1):
            e0.record()
            assert lib_near.FlowProjectionLayer_gpu_forward(tn, cn, on, fill) == 0
            e1.record()
        blk_n = last_block()
        if serial:
            torch.cuda.synchronize()
        with torch.cuda.stream(s2):
            e2.record()
            assert lib_far.FlowProjectionLayer_gpu_forward(tf, cf, of, fill) == 0
            e3.record()
        blk_f = last_block()
        outs.append((on, cn, of, cf))
        if it < 2:
            torch.cuda.synchronize()
            notes.append({"near_stream": s1.cuda_stream, "far_stream": s2.cuda_stream, "near_block": blk_n, "far_block": blk_f, "near_call_us": round(e0.elapsed_time(e1) * 1e3, 1),
                          "far_call_us": round(e2.elapsed_time(e3) * 1e3, 1),
                          "far_starts_after_near_starts_us": round(e0.elapsed_time(e2) * 1e3, 1),
                          "near_out": [on.data_ptr(), on.numel() * 4], "far_out": [of.data_ptr(), of.numel() * 4]})
    torch.cuda.synchronize()
    del LAST_NOTES[:]
    LAST_NOTES.extend(notes)
    bad_n = bad_f = 0
    for it, (on, cn, of, cf) in enumerate(outs):
        for name, o, c, (wo, wc) in (("near", on, cn, want_n), ("far", of, cf, want_f)):
            go, gc = o.cpu().numpy(), c.cpu().numpy()
            ok = np.all(np.abs(go - wo) <= ATOL * np.maximum(1.0, np.abs(wo))) and np.array_equal(gc, wc)
            if not ok:
                if name == "near":
                    bad_n += 1
                else:
                    bad_f += 1
                if log is not None and len(log) < 6:
                    log.append({"run": run, "iteration": it, "stream": name, "calls": notes[it] if it < len(notes) else None,
                                **describe(go, gc, wo, wc, H, W)})
    return bad_n, bad_f


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=12)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "far_spill_streams.txt"))
    ap.add_argument("--product", default="", help="label: run the test's loop on whatever lib/libmemc_hip.so is (a variant "
                    "copied over it by the session script), with the diagnostics, instead of the measurement build's arms")
    a = ap.parse_args()
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    rng = np.random.default_rng(31)
    near = synth.np_flow(rng, 2, 64, 128, "smooth", 3.0)
    far = synth.np_flow(rng, 2, 64, 128, "iid", 40.0)
    rng2 = np.random.default_rng(32)
    near_b = synth.np_flow(rng2, 4, 256, 512, "smooth", 3.0)     # a larger pair: more tiles, more time in flight
    far_b = synth.np_flow(rng2, 4, 256, 512, "iid", 40.0)
    want = {}
    for key, arr in (("near", near), ("far", far), ("near_b", near_b), ("far_b", far_b)):
        for fill in (0, 1):
            o, c = O.flow_projection_forward(arr, fill)
            want[key, fill] = (o, c)
    t = {"near": T(near), "far": T(far), "near_b": T(near_b), "far_b": T(far_b)}

    class Prod(object):
        FlowProjectionLayer_gpu_forward = staticmethod(PL.FlowProjectionLayer_gpu_forward)

    # (label, projection variant of the measurement build, near library, far library, near key, far key, fill, serial)
    arms = [
        ("A  product kernels (measurement build, variant -1), fill 1", -1, ML, ML, "near", "far", 1, False),
        ("B  spilling far kernel, three per CU (-44), fill 1  [round 4's failing build]", -44, ML, ML, "near", "far", 1, False),
        ("C  spilling far kernel on the product's grid (-45), fill 1", -45, ML, ML, "near", "far", 1, False),
        ("D  -44, fill 0 (no proj_fill_pending, no fill epilogue)", -44, ML, ML, "near", "far", 0, False),
        ("E  -44, the two calls serialised by a device synchronisation", -44, ML, ML, "near", "far", 1, True),
        ("F  -44, BOTH streams near (nobody's far kernel works)", -44, ML, ML, "near", "near", 1, False),
        ("G  -44, BOTH streams far", -44, ML, ML, "far", "far", 1, False),
        ("H  -44 on the far stream only; the near stream on the PRODUCT library (no scratch dispatch on it)", -44, Prod, ML, "near", "far", 1, False),
        ("I  -44 on the near stream only (its idle dispatch asks for scratch); far stream on the product library", -44, ML, Prod, "near", "far", 1, False),
        ("J  -44, 4 x 256 x 512 images", -44, ML, ML, "near_b", "far_b", 1, False),
        ("K  product kernels, 4 x 256 x 512 images", -1, ML, ML, "near_b", "far_b", 1, False),
    ]
    fixed = None
    if a.product:
        arms = [("%s: near || far, fill 1, NEW streams every run" % a.product, None, Prod, Prod, "near", "far", 1, False),
                ("%s: near || far, fill 1, the SAME two streams in every run" % a.product, "same", Prod, Prod, "near", "far", 1, False),
                ("%s: near || far, fill 0, new streams" % a.product, None, Prod, Prod, "near", "far", 0, False),
                ("%s: near || far, serialised, new streams" % a.product, None, Prod, Prod, "near", "far", 1, True),
                ("%s: near || near, new streams" % a.product, None, Prod, Prod, "near", "near", 1, False)]
    lines = []
    for label, variant, ln, lf, kn, kf, fill, serial in arms:
        if variant == "same":
            fixed = (torch.cuda.Stream(), torch.cuda.Stream())
        elif variant is not None:
            M.set_variant("projection", variant)
        log = []
        first_notes = []
        tot_n = tot_f = runs_bad = 0
        for r in range(a.rounds):
            bn, bf = one_run(ln, lf, t[kn], t[kf], want[kn, fill], want[kf, fill], fill, 20, serial, log,
                             streams=fixed if variant == "same" else None, run=r)
            tot_n += bn
            tot_f += bf
            if r < 2 and LAST_NOTES:
                first_notes.append(LAST_NOTES[0])
            runs_bad += 1 if (bn or bf) else 0
        line = "%-100s runs with a wrong result: %2d of %d   wrong iterations: stream 1 (%s) %3d, stream 2 (%s) %3d of %d" % (
            label, runs_bad, a.rounds, kn, tot_n, kf, tot_f, 20 * a.rounds)
        print(line, flush=True)
        lines.append(line)
        for e in first_notes[:2]:
            lines.append("      first iteration of a run: " + json.dumps(e))
            print("      first iteration of a run: " + json.dumps(e), flush=True)
        for e in log:
            lines.append("      " + json.dumps(e))
            print("      " + json.dumps(e), flush=True)
    M.set_variant("projection", -1)
    with open(a.out, "a") as f:
        f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
