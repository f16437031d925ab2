#!/usr/bin/env python
"""Builds (hipcc, in-tree) and runs tools/probes/skeleton_walks.hip on cuda:0: the adaptive warp's I/O skeleton by walk order."""
import ctypes, os, statistics, subprocess, sys
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "libskeleton_walks.so")


def build():
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                    "-o", SO, os.path.join(HERE, "skeleton_walks.hip")], check=True)


def main():
    if not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(os.path.join(HERE, "skeleton_walks.hip")):
        build()
    if not torch.cuda.is_available():
        print("built", SO); return
    lib = ctypes.CDLL(SO)
    dev = torch.device("cuda:0")
    B, H, W = 32, 720, 1280
    x = torch.rand(B, 3, H, W, device=dev); f = torch.randn(B, 2, H, W, device=dev)
    k = torch.rand(B, 16, H, W, device=dev); o = torch.zeros_like(x)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    nbytes = B * H * W * 96
    walks = {0: "raster, chunk per XCD", 1: "raster, b = tile", 2: "strips (product)", 3: "column classes", 4: "stripes of 2"}
    cases = []
    for lx in (16, 32, 64):
        for walk in (0, 1, 2, 3, 4):
            for G in ((1, 2, 5, 10, 20, 40) if walk == 3 else (0,)):
                cases.append((lx, 0, walk, G))
    cases += [(16, 1, 2, 0), (16, 2, 2, 0), (16, 2, 1, 0), (64, 2, 1, 0)]
    if len(sys.argv) > 2 and sys.argv[2] == "policies":      # the stores' cache-policy bits, strips, 64 x 16 tiles
        cases = [(16, 0, 2, 0)] + [(16, 8 + bits, 2, 0) for bits in range(8)]
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    if len(sys.argv) > 2 and sys.argv[2] == "loads":
        return loads(lib, st, B, H, W, nbytes, rounds)
    if len(sys.argv) > 2 and sys.argv[2] == "persistent":
        return persistent(lib, st, P, B, H, W, x, f, k, o, nbytes, rounds)
    if len(sys.argv) > 2 and sys.argv[2] == "phased":
        return phased(lib, st, P, B, H, W, x, f, k, o, nbytes, rounds)
    ts = {c: [] for c in cases}
    bad = set()
    for r in range(rounds):
        for c in cases:
            if c in bad:
                continue
            lx, wr, walk, G = c
            call = lambda: lib.probe_skeleton_walk(st, lx, wr, walk, G, B, H, W, P(x), P(f), P(k), P(o))
            if call() != 0:
                bad.add(c); continue
            call(); call()
            for _ in range(8):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); call(); b.record(); b.synchronize()
                ts[c].append(a.elapsed_time(b) * 1e-3)
    for c in cases:
        if c in bad:
            continue
        lx, wr, walk, G = c
        t = statistics.median(ts[c])
        moved = nbytes if wr != 2 else nbytes * 84 // 96
        wname = ("nt stores", "plain st.", "no stores")[wr] if wr < 8 else "st" + "".join(
            n for bit, n in ((1, " sc0"), (2, " sc1"), (4, " nt")) if (wr - 8) & bit)
        print("tile %3dx%-2d %-13s %-22s %-5s %8.1f us  %7.1f GB/s  %5.1f%% of 8 TB/s" % (
            4 * lx, 256 // lx, wname, walks[walk], ("G=%d" % G) if walk == 3 else "",
            t * 1e6, moved / t / 1e9, 100 * moved / t / 8e12), flush=True)


def loads(lib, st, B, H, W, nbytes, rounds):
    """every cache policy of the stream LOADS, on contiguous tensors and on rows padded by 64 floats / planes padded by 64 floats"""
    dev = torch.device("cuda:0")
    lib.probe_skeleton_loads.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int64,
                                         ctypes.c_int64] + [ctypes.c_void_p] * 4
    layouts = {"contiguous": (W * H, W), "rows + 256 B": (H * (W + 64), W + 64), "planes + 256 B": (W * H + 64, W)}
    bufs = {}
    for name, (plane, rowp) in layouts.items():
        bufs[name] = tuple(torch.rand(B * c * plane + 64, device=dev) for c in (3, 2, 16)) + (torch.zeros(B * 3 * plane + 64, device=dev),)
    cases = [(name, bits) for name in layouts for bits in range(8)]
    ts = {c: [] for c in cases}
    for r in range(rounds):
        for c in cases:
            name, bits = c
            plane, rowp = layouts[name]
            x, f, k, o = bufs[name]
            call = lambda: lib.probe_skeleton_loads(st, bits, B, H, W, plane, rowp, x.data_ptr(), f.data_ptr(), k.data_ptr(), o.data_ptr())
            assert call() == 0
            call()
            for _ in range(6):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); call(); b.record(); b.synchronize()
                ts[c].append(a.elapsed_time(b) * 1e-3)
    for c in cases:
        t = statistics.median(ts[c])
        pol = "ld" + "".join(n for bit, n in ((1, " sc0"), (2, " sc1"), (4, " nt")) if c[1] & bit)
        print("64x16 strips, %-15s loads %-14s %8.1f us  %7.1f GB/s  %5.1f%% of 8 TB/s" % (c[0], pol, t * 1e6, nbytes / t / 1e9, 100 * nbytes / t / 8e12), flush=True)


def persistent(lib, st, P, B, H, W, x, f, k, o, nbytes, rounds):
    """persistent workgroups that hold up to NP tiles' results for the write window (skeleton_persistent)"""
    cases = [("plain", 0, 0, 0, 0)]
    pause = len(sys.argv) > 3 and sys.argv[3] == "pause"
    for occ in (3, 4, 6, 8):
        cases.append(("pers", 0, occ, 1000, 100))
        for np_ in ((102, 104) if pause else (1, 2, 3, 4)):
            for per, win in ((500, 80), (1000, 130), (1000, 200), (1500, 250), (2000, 260), (2000, 400), (3000, 500), (4000, 520)):
                cases.append(("pers", np_, occ, per, win))
    ts = {c: [] for c in cases}
    for r in range(rounds):
        for c in cases:
            kind, np_, occ, per, win = c
            if kind == "plain":
                call = lambda: lib.probe_skeleton_walk(st, 16, 0, 2, 0, B, H, W, P(x), P(f), P(k), P(o))
            else:
                call = lambda: lib.probe_skeleton_persistent(st, np_, occ, per, win, B, H, W, P(x), P(f), P(k), P(o))
            assert call() == 0, c
            call()
            for _ in range(5):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); call(); b.record(); b.synchronize()
                ts[c].append(a.elapsed_time(b) * 1e-3)
    for c in cases:
        kind, np_, occ, per, win = c
        t = statistics.median(ts[c])
        what = "one workgroup per tile, stores as they come" if kind == "plain" else (
            "persistent %d/CU, stores as they come" % occ if np_ == 0 else
            "persistent %d/CU, holds %d tiles%s, window %.1f of %.0f us" % (
                occ, np_ % 100, " + pauses reads" if np_ > 100 else "", win / 100.0, per / 100.0))
        print("64x16 strips: %-58s %8.1f us  %7.1f GB/s  %5.1f%% of 8 TB/s" % (what, t * 1e6, nbytes / t / 1e9, 100 * nbytes / t / 8e12), flush=True)


def phased(lib, st, P, B, H, W, x, f, k, o, nbytes, rounds):
    """stores held until a chip-wide write window of the 100 MHz clock (skeleton_phased); (0, 0) = the plain skeleton"""
    cases = [(0, 0, 0)] + [(per, win, mode) for mode in (0, 1, 2) for per in (500, 800, 1000, 1200, 1500, 2000)
                           for win in (per // 8, per // 5, per // 3)]
    if len(sys.argv) > 3 and sys.argv[3] == "wbl2":
        cases = [(0, 0, 0), (1000, 100, 4), (1000, 120, 0)] + [(per, win, 3) for per in (200, 500, 1000, 2000, 4000, 8000)
                                                             for win in (per // 50 + 1, per // 16, per // 8)]
    ts = {c: [] for c in cases}
    for r in range(rounds):
        for c in cases:
            per, win, mode = c
            if per == 0:
                call = lambda: lib.probe_skeleton_walk(st, 16, 0, 2, 0, B, H, W, P(x), P(f), P(k), P(o))
            else:
                call = lambda: lib.probe_skeleton_phased(st, 16 + 256 * mode, 2, per, win, B, H, W, P(x), P(f), P(k), P(o))
            assert call() == 0
            call()
            for _ in range(6):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); call(); b.record(); b.synchronize()
                ts[c].append(a.elapsed_time(b) * 1e-3)
    for c in cases:
        t = statistics.median(ts[c])
        print("64x16 strips, %-62s %8.1f us  %7.1f GB/s  %5.1f%% of 8 TB/s" % (
            "stores as they come" if c[0] == 0 else "%s in the last %.1f us of every %.0f us" % (
                ("stores", "workgroup starts", "starts (stores half a period later)", "cached stores, L2 write-back requested",
                 "cached stores, no request (control):")[c[2]], c[1] / 100.0, c[0] / 100.0),
            t * 1e6, nbytes / t / 1e9, 100 * nbytes / t / 8e12), flush=True)


if __name__ == "__main__":
    main()
