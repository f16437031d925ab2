#!/usr/bin/env python
"""tools/kernel_resources.py -- per-kernel registers / LDS / private scratch of the PRODUCT build, from the compiler's own
remarks (hipcc -Rpass-analysis=kernel-resource-usage; cross-compiles without a GPU).

    python tools/kernel_resources.py            # table of every kernel of memc-net_amd/csrc/*.hip
    python tools/kernel_resources.py --scratch  # only the kernels that use private scratch (exit 1 if there is one)
    python tools/kernel_resources.py --measure  # the measurement build (-DMEMC_MEASURE: arms included)

Used by tests/test_abi.py::test_no_product_kernel_uses_private_scratch: round 4 saw a wrong result on a stream that ran
next to an experimental kernel that spilled; whatever the cause, a product kernel that spills is
a performance bug first, so none may."""
import argparse
import concurrent.futures
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "memc-net_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-munsafe-fp-atomics",
         "-I" + os.path.join(ROOT, "include"), "-Rpass-analysis=kernel-resource-usage", "-c", "-o", "/dev/null"]
SOURCES = ["filter_interpolation.hip", "fi_bwd_c3.hip", "fi_bwd_cn.hip", "interpolation.hip", "flow_projection.hip",
           "flow_prologue.hip", "calibration.hip"]

_KEYS = {"Function Name": "name", "TotalSGPRs": "sgprs", "VGPRs": "vgprs", "AGPRs": "agprs",
         "ScratchSize [bytes/lane]": "scratch", "Occupancy [waves/SIMD]": "occupancy",
         "LDS Size [bytes/block]": "lds", "Dynamic Stack": "dynstack"}


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), text=True,
                             stdout=subprocess.PIPE, check=True).stdout.split("\n")
        return [re.sub(r"^void ", "", re.sub(r"\(.*$", "", o)) for o in out[:len(names)]]
    except Exception:
        return names


def resources_of(src, defs=()):
    r = subprocess.run([HIPCC] + FLAGS + list(defs) + [src], cwd=CSRC, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed on %s:\n%s" % (src, r.stdout[-2000:]))
    kernels, cur = [], None
    for line in r.stdout.split("\n"):
        m = re.search(r"remark: [^:]+:\d+:\d+:\s+(.*?):\s+(\S+)\s+\[-Rpass-analysis", line) or \
            re.search(r"remark:\s+(.*?):\s+(\S+)\s+\[-Rpass-analysis", line)
        if not m:
            continue
        key, val = m.group(1).strip(), m.group(2)
        if key == "Function Name":
            cur = {"file": src, "name": val}
            kernels.append(cur)
        elif cur is not None and key in _KEYS:
            cur[_KEYS[key]] = val
    names = demangle([k["name"] for k in kernels])
    for k, n in zip(kernels, names):
        k["name"] = n
    return kernels


def all_resources(measure=False):
    defs = ("-DMEMC_MEASURE",) if measure else ()
    srcs = SOURCES + (["arms/fi_bwd_c3_arms.hip"] if measure else [])
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        return [k for ks in ex.map(lambda s: resources_of(s, defs), srcs) for k in ks]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scratch", action="store_true")
    ap.add_argument("--measure", action="store_true")
    a = ap.parse_args()
    ks = all_resources(a.measure)
    if a.scratch:
        ks = [k for k in ks if int(k.get("scratch", "0")) > 0]
    print("%-28s %-60s %5s %5s %5s %7s %4s %7s" % ("file", "kernel", "VGPR", "AGPR", "SGPR", "scratch", "occ", "LDS"))
    for k in ks:
        print("%-28s %-60s %5s %5s %5s %7s %4s %7s" % (k["file"], k["name"][:60], k.get("vgprs", "?"), k.get("agprs", "?"),
                                                      k.get("sgprs", "?"), k.get("scratch", "?"), k.get("occupancy", "?"),
                                                      k.get("lds", "?")))
    if a.scratch and ks:
        sys.exit(1)


if __name__ == "__main__":
    main()
