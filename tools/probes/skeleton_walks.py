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


if __name__ == "__main__":
    main()
