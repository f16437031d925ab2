"""Prints the XCC_ID seen by workgroup b for a few grid shapes: is it b % 8?  (tools/probes/xcc_probe.hip)"""
import ctypes, os, collections, torch
HERE = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(HERE, "libxcc_probe.so"))
dev = torch.device("cuda:0")
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for (n, threads, lds) in ((64, 256, 0), (28800, 256, 49216), (28800, 256, 0), (2048, 64, 0)):
    out = torch.full((n,), -1, dtype=torch.int32, device=dev)
    assert lib.probe_xcc(st, ctypes.c_void_p(out.data_ptr()), n, threads, lds) == 0
    torch.cuda.synchronize()
    v = out.cpu().tolist()
    raw = sorted(set(v))
    ids = [x & 0xF for x in v]
    match = sum(1 for b, x in enumerate(ids) if x == b % 8)
    table = collections.Counter((b % 8, x) for b, x in enumerate(ids))
    print("grid %d x %d threads, %d B LDS: raw values %s; xcc == b %% 8 for %d of %d" % (n, threads, lds, [hex(r) for r in raw[:12]], match, n))
    print("   first 24:", ids[:24])
    if match != n:
        print("   (b%8, xcc) counts:", sorted(table.items())[:24])
