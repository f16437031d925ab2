#!/bin/bash
# Round 5, GPU session 22: lanes without a hole store BEFORE the fill epilogue's first barrier: projection tests, A/B against the
# build before on the benchmark's flow, i.i.d. flow is covered by the tests; pans 8 / 40, flow x 2.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05_s22
mkdir -p "$OUT"
cd "$REPO"
LIB=memc-net_amd/lib/libmemc_hip.so
OLD=tools/probes/variants/libmemc_hip_before_early_store.so
timeout 900 python -m pytest tests/test_gpu_workspace_and_streams.py tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py tests/test_gpu_reference.py -q -m gpu -k "workspace or graph or streams or thread or projection or Projection or pan or hole or far or ragged or multiples or config3 or stalled" 2>&1 | tail -3 | tee $OUT/pytest_proj.log
for ARGS in "--pan 0" "--pan 8" "--pan 40" "--scale 2"; do
  echo "== $ARGS" | tee -a $OUT/ab.txt
  timeout 300 python tools/ab_libs.py $OLD $LIB --op proj,proj_fill,depth_fill --rounds 8 $ARGS 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab.txt
done
