#!/bin/bash
# Round 6, GPU session 19: the headline kernel's walks, unprofiled, one process (product / stripes 2 / stripes 4 / row-major
# chunks / hardware order), next to the I/O skeleton of the operator by tile shape on the same box.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r06_s19
mkdir -p "$OUT"
cd "$REPO"
rm -f tools/probes/libio_skeleton.so
timeout 300 python tools/probes/run_probe.py 2>&1 | grep -v amdgpu.ids | tee "$OUT/io_skeleton.txt"
for r in 1 2; do
timeout 600 python tools/ab_variants.py --op fi_fwd --variants=-1,15,16,17,8 --cases fi_fwd --flows smooth,iid --rounds 8 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/fi_fwd_walks_ab.txt"
done
