#!/usr/bin/env python
"""Builds (hipcc, in-tree) and runs tools/probes/io_skeleton.hip on cuda:0; prints achieved bandwidth."""
import ctypes, os, statistics, subprocess, sys
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "libio_skeleton.so")


def build():
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                    "-o", SO, os.path.join(HERE, "io_skeleton.hip")], check=True)


def timeit(fn, iters=15, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize()
        ts.append(a.elapsed_time(b) * 1e-3)
    return statistics.median(ts)


def main():
    if not os.path.exists(SO):
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
    names = {0: "64x16 nt xcd", 1: "32x32 nt xcd", 2: "256x4 nt xcd", 3: "128x8 nt xcd", 4: "64x16 cached xcd",
             5: "64x16 nt blockIdx-order", 6: "256x4 nt blockIdx-order", 7: "64x16 nt xcd minw2", 8: "256x4 cached blockIdx-order"}
    for v in range(9):
        t = timeit(lambda: lib.probe_skeleton(st, v, B, H, W, P(x), P(f), P(k), P(o)))
        print("skeleton %-28s %8.1f us  %7.1f GB/s  %5.1f%% of 8 TB/s" % (names[v], t * 1e6, nbytes / t / 1e9, 100 * nbytes / t / 8e12), flush=True)
    n4 = nbytes // 32
    a = torch.rand(n4 * 4, device=dev); b = torch.empty_like(a)
    lib.probe_copy.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]
    for blocks in (2048, 8192, 65536):
        for v, nm in ((0, "copy cached"), (1, "copy nt"), (2, "read-only nt")):
            t = timeit(lambda: lib.probe_copy(st, v, blocks, P(a), P(b), n4))
            moved = n4 * 16 * (1 if v == 2 else 2)
            print("%-14s blocks=%-6d %8.1f us  %7.1f GB/s" % (nm, blocks, t * 1e6, moved / t / 1e9), flush=True)


def atomics_main():
    lib = ctypes.CDLL(SO)
    dev = torch.device("cuda:0")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    lib.probe_atomics.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int]
    n = 32 * 720 * 1280 * 3
    d = torch.zeros(n, device=dev)
    for rep, shift in ((1, 0), (4, 0), (4, 1), (4, 1280), (2, 1)):
        t = timeit(lambda: lib.probe_atomics(st, 0, 16384, ctypes.c_void_p(d.data_ptr()), n, rep, shift))
        print("atomics n=%d rep=%d shift=%-5d %9.1f us  %7.2f G atomics/s" % (n, rep, shift, t * 1e6, n * rep / t / 1e9), flush=True)
    # is the coalesced rate the atomic units' or HBM's?  The same pattern over working sets that stay in L2 / MALL
    for n_small, rep in ((1 << 20, 64), (1 << 23, 16), (1 << 25, 4)):
        t = timeit(lambda: lib.probe_atomics(st, 0, 16384, ctypes.c_void_p(d.data_ptr()), n_small, rep, 0))
        print("atomics n=%d (%.0f MB) rep=%d shift=0 %9.1f us  %7.2f G atomics/s" % (
            n_small, n_small * 4 / 1e6, rep, t * 1e6, n_small * rep / t / 1e9), flush=True)
    t = timeit(lambda: lib.probe_atomics(st, 1, 16384, ctypes.c_void_p(d.data_ptr()), n, 1, 0))
    print("plain float4 += over the same %d elements %9.1f us  %7.1f GB/s (read+write)" % (n, t * 1e6, n * 8 / t / 1e9))


def lds_main():
    lib = ctypes.CDLL(SO)
    dev = torch.device("cuda:0")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    sink = torch.zeros(4, device=dev)
    blocks, rep = 256 * 8, 4096
    for mode, nm in ((3, "ds_add_f64"), (4, "ds_add_u64"), (0, "ds_add_f32"), (1, "ds_add_u32")):
        for stride, rstep in ((1, 256), (4, 1), (4, 97)):
            t = timeit(lambda: lib.probe_lds_atomics(st, blocks, ctypes.c_void_p(sink.data_ptr()), rep, stride, rstep, mode), iters=5, warmup=2)
            ops = blocks * 256 * rep
            print("%-11s stride=%-3d rstep=%-4d %8.1f us  %8.1f G lane-ops/s  (%.2f per clk per CU at 2.4 GHz)" % (
                nm, stride, rstep, t * 1e6, ops / t / 1e9, ops / t / 256 / 2.4e9), flush=True)


def lds_pattern_main():
    lib = ctypes.CDLL(SO)
    dev = torch.device("cuda:0")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    sink = torch.zeros(4, device=dev)
    blocks, rep = 256 * 8, 4096
    names = {0: "consecutive, 64 lanes", 1: "consecutive, 16 of 64 lanes", 2: "consecutive, 8 of 64 lanes",
             3: "kernel layout, no noise", 4: "kernel layout, noise 0..4 cells", 5: "pitch 96, no swizzle, noise",
             6: "kernel layout, 32 of 64 lanes", 7: "rows interleaved in one line"}
    for pat in sorted(names):
        t = timeit(lambda: lib.probe_lds_pattern(st, blocks, ctypes.c_void_p(sink.data_ptr()), rep, pat), iters=5, warmup=2)
        instr = blocks * 4 * rep                       # wave-instructions
        print("ds_add_f64 %-34s %8.1f us  %6.1f clk per wave-instruction per CU (2.4 GHz)" % (
            names[pat], t * 1e6, t * 2.4e9 * 256 / instr), flush=True)


def copy_only():
    """a few launches of the float4 copy of known size (calibration source for tools/pmc_traffic.py)"""
    lib = ctypes.CDLL(SO)
    dev = torch.device("cuda:0")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    lib.probe_copy.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]
    n4 = 32 * 720 * 1280 * 96 // 32
    a = torch.rand(n4 * 4, device=dev); b = torch.empty_like(a)
    for _ in range(6):
        lib.probe_copy(st, 1, 65536, ctypes.c_void_p(a.data_ptr()), ctypes.c_void_p(b.data_ptr()), n4)
    torch.cuda.synchronize()


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "copyonly":
        copy_only()
    elif len(sys.argv) > 1 and sys.argv[1] == "build":
        build()
    elif len(sys.argv) > 1 and sys.argv[1] == "ldspat":
        lds_pattern_main()
    elif len(sys.argv) > 1 and sys.argv[1] == "lds":
        lds_main()
    elif len(sys.argv) > 1 and sys.argv[1] == "atomics":
        atomics_main()
    else:
        main()
