#!/bin/bash
# Round 4, GPU session 1: the new projection set (proj_owner5 + mask filler) -- parity, then A/B against round 3's set in one process,
# phase clocks, SQ counters incl. the scalar pipe, and the in-library streaming calibration.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04_s1
mkdir -p "$OUT"
cd "$REPO"
echo "== parity: projection tests"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference.py tests/test_gpu_baseline_configs.py -m gpu -q -x -k "proj or hole or pan or config3 or fill" 2>&1 | tail -8 | tee "$OUT/pytest_proj.log"
echo "== stress"; timeout 600 python tools/stress_projection.py 40 2>&1 | tail -6 | tee "$OUT/stress.log"
echo "== A/B in one process: -1 = proj_owner5 + mask filler, -40 = round 3's set"
timeout 600 python tools/ab_variants.py --op projection --variants=-1,-40 --flows smooth,iid 2>&1 | grep -v amdgpu.ids | tee "$OUT/ab_proj.txt"
echo "== phase clocks"
timeout 300 python tools/trace_kernel.py proj5 2>&1 | grep -v amdgpu.ids | tee "$OUT/proj5_trace.txt"
echo "== calibration"
timeout 300 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee "$OUT/calibration.txt"
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "memc-net_amd"))
import torch, bench
import my_package._ext.my_lib as L
dev = torch.device("cuda:0")
a = torch.empty(2831155200 // 8, dtype=torch.float32, device=dev).normal_(); b = torch.empty_like(a)
for _ in range(200): b.copy_(a)
for rep in range(3):
    r = bench.copy_calibration(L, dev, 2831155200)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
    for s0, s1 in ev:
        s0.record(); b.copy_(a); s1.record()
    torch.cuda.synchronize()
    ts = sorted(s0.elapsed_time(s1) for s0, s1 in ev)
    print("rep", rep, {k: round(v / 1e9, 1) for k, v in r.items()}, "torch copy_ GB/s", round(2 * a.numel() * 4 / ts[10] / 1e6, 1))
PY
echo "== SQ counters of proj_owner5 (three passes)"
bash tools/pmc_sq.sh r04_s1/sq proj "proj_owner5<false" 2>&1 | tail -45 | tee "$OUT/proj5_sq.txt"
ls "$OUT"
