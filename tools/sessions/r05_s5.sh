#!/bin/bash
# Round 5, GPU session 5: (a) does the runtime's stream-ordered pool hand stream B a block stream A has freed (in stream order)
# while A's kernels still use it?  + the failing variant: did its blocks come from the cache entry or from per-call
# allocations; (b) ragged widths on the owner kernels: tests + what they cost; (c) the suite's projection part.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05_s5
mkdir -p "$OUT"
cd "$REPO"
LIB=memc-net_amd/lib/libmemc_hip.so
timeout 120 tools/probes/scratch_streams 0 2>&1 | tee $OUT/pool_probe.txt
cp $LIB /tmp/libmemc_hip.current.so
cp tools/probes/variants/libmemc_hip_farArm_oldScratch.so $LIB
timeout 300 python tools/probes/far_spill_streams.py --rounds 3 --product farArm_oldScratch --out $OUT/variants.txt 2>&1 | grep -v amdgpu.ids | cut -c1-700 | head -8
cp /tmp/libmemc_hip.current.so $LIB
echo "== tests (current tree)"
timeout 900 python -m pytest tests/test_gpu_workspace_and_streams.py tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py tests/test_gpu_reference.py -q -m gpu -k "workspace or graph or streams or thread or projection or Projection or pan or hole or far or unaligned or documented or ragged or multiples or config3" 2>&1 | tail -8 | tee $OUT/pytest_proj.log
echo "== slow paths"
timeout 400 python tools/probes/slow_paths.py 2>&1 | grep -v amdgpu.ids | tee $OUT/slow_paths.txt
