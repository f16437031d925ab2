#!/bin/bash
# Round 4, GPU session 22: the far branch of proj_owner5 reduces its landing box per wave before the LDS atomics (a pan ran the
# kernel at 1.1 ms) -- parity, kernel trace of the panned projection, both motion sweeps, the fast path's burst timing.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04_s22
mkdir -p "$OUT"
cd "$REPO"
echo "== parity"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference.py tests/test_gpu_baseline_configs.py -m gpu -q -k "proj or hole or pan or config3 or fill or unusual or far or stream" 2>&1 | tail -5 | tee "$OUT/pytest.log"
echo "== stress"; timeout 600 python tools/stress_projection.py 40 2>&1 | tail -1 | tee "$OUT/stress.log"
echo "== kernel trace, pan 40, fill 1"
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/t1 -o r -- python $REPO/tools/probes/proj_far_load.py 1 40 1 > $OUT/t1.log 2>&1
 python $REPO/tools/prof_summary.py stats $OUT/t1/r_results.db 2>/dev/null | head -5 | tee $OUT/pan40_fill1.txt; rm -rf $OUT/t1)
echo "== motion sweep, projection"
timeout 300 python tools/probes/proj_motion_sweep.py 2>&1 | grep -v amdgpu.ids | tee "$OUT/motion_sweep.txt"
echo "== motion sweep, all operators"
timeout 300 python tools/probes/motion_sweep_all.py 2>&1 | grep -v amdgpu.ids | tee "$OUT/motion_sweep_all.txt"
echo "== burst timing"
timeout 300 python tools/probes/proj_burst.py 2>&1 | grep -v amdgpu.ids | tee "$OUT/burst.txt"
