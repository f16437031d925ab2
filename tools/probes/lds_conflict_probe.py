#!/usr/bin/env python
"""Builds and runs tools/probes/lds_conflict_probe.hip.  Under rocprofv3:
    rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CU_CYCLES --kernel-trace -d out -o r -- python tools/probes/lds_conflict_probe.py
then `python tools/prof_summary.py pmc out/r_results.db --match lds_probe`."""
import ctypes, os, subprocess, sys
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "liblds_conflict_probe.so")
if not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(os.path.join(HERE, "lds_conflict_probe.hip")):
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-o", SO,
                    os.path.join(HERE, "lds_conflict_probe.hip")], check=True)
if not torch.cuda.is_available():
    print("built", SO); sys.exit(0)
lib = ctypes.CDLL(SO)
lib.lds_probe_run.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
out = torch.zeros(2048 * 256, device="cuda")
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
sums = {}
for p in range(20):
    for _ in range(3):
        assert lib.lds_probe_run(st, p, ctypes.c_void_p(out.data_ptr()), 2048, 64) == 0
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); lib.lds_probe_run(st, p, ctypes.c_void_p(out.data_ptr()), 2048, 64); b.record(); b.synchronize()
    sums[p] = float(out.double().sum())
    print("pattern %d: %.1f us  (sum of results %.6g)" % (p, a.elapsed_time(b) * 1e3, sums[p]), flush=True)
for p in (14, 15, 16):
    print("unaligned ds_read_b128, pattern %d against %d: %s" % (p, p - 3, "same values" if sums[p] == sums[p - 3] else "DIFFERENT"))
