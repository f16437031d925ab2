#!/bin/bash
# One GPU-box session: smoke, parity tests (with the record of observed errors), operator sweep, baselines, headline bench,
# shard sizes, model bench, rocprofv3 kernel traces and the PMC passes (FETCH_SIZE / WRITE_SIZE need separate passes: TCC has
# 4 slots, they cost 3 + 2).
# Usage (from the repo root, via gpurun):  bash tools/gpu_session.sh <tag> [quick]
set -u
TAG=${1:-r06}
QUICK=${2:-}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$REPO"
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== smoke";   timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee "$OUT/smoke.log"
echo "== pytest";  timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -15 | tee "$OUT/pytest_gpu.log"
cp gpurun_out/parity_errors.json "$OUT/parity_errors.json" 2>/dev/null
echo "== sweep (product library)";   timeout 900 python tools/bench_ops.py ${QUICK:+--quick} --json "$OUT/bench_ops.json" 2>&1 | grep -v amdgpu.ids | tee "$OUT/bench_ops.log"
echo "== baselines (reference kernels on this GPU, CPU oracle)"; timeout 900 python tests/bench_baselines.py --json "$OUT/baselines.json" 2>&1 | grep -v amdgpu.ids | tee "$OUT/baselines.log"
echo "== bench";   timeout 900 python bench.py 2>&1 | tail -3 | tee "$OUT/bench.log"
echo "== bench at the driver's arguments (--steps 20 --warmup 5), whole line"
timeout 900 python bench.py --steps 20 --warmup 5 2>&1 | tail -1 | tee "$OUT/bench_driver_args.log" | cut -c1-400
echo "== bench, strong-scaling shard sizes on one GPU (what rank 0 of an N-GPU run executes: batch 32 / N, rotating input sets), at the driver's 20 steps and at 300"
for b in 32 16 8 4; do for st in 20 300; do timeout 300 python bench.py --batch $b --steps $st --warmup 5 --no-cpu-baseline --no-secondary 2>&1 | tail -1 | tee -a "$OUT/bench_shards.log" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']; r=d['roofline']
print('batch %2d steps %3d %-9s value %9.1f ms/step %.4f gpu_us/step %.2f enqueue_us %.1f fixed_us %.1f kernel_us %.2f first %s windows %s' % (c['batch_per_gpu'], d['steps'], c['launch'], d['value'], d['ms_per_step'], c['window_gpu_us_per_step'], c['window_host_enqueue_us'], c['window_fixed_cost_us'], r['avg_launch_us'], r['first_launches_us'], c['window_ms_min_max']))" | tee -a "$OUT/bench_shards_summary.txt"; done; done
echo "== the N > 1 code path on this one GPU: two ranks sharing it over gloo (PLUMBING, not a measurement), 20 and 300 steps"
echo "== the collective layer on a one-rank RCCL communicator (backend nccl on this one GPU)"
timeout 600 python bench.py --dist-single --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>&1 | grep '^{"metric"' | tee "$OUT/bench_dist_single.log" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('one-rank RCCL communicator:', json.dumps(d['dist']))" | tee -a "$OUT/bench_shards_summary.txt"
T0=$(date +%s.%N)
for st in 20 300; do timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29641 bench.py --gpus 2 --share-gpu --batch 8 --steps $st --warmup 5 2>&1 | grep '^{"metric"' | tee -a "$OUT/bench_share_gpu.log" | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('2 ranks on one GPU: steps %3d ms/step %.4f barrier_us %s per-rank ms/step %s' % (d['steps'], d['ms_per_step'], d['dist']['barrier_us'], d['dist']['per_rank_ms_per_step']))" | tee -a "$OUT/bench_shards_summary.txt"; done
python -c "import sys; print('the two 2-rank runs (torch.distributed.run start-up, two processes, gloo) took %.1f s end to end together' % (float(sys.argv[2]) - float(sys.argv[1])))" $T0 $(date +%s.%N) | tee -a "$OUT/bench_shards_summary.txt"
echo "== model"; timeout 600 python tools/bench_model.py --json "$OUT/bench_model.json" 2>&1 | tail -2 | tee "$OUT/bench_model.log"
echo "== projection: phase clocks of the owner kernel (measurement build), A/B against round 3's set in one process"
timeout 300 python tools/trace_kernel.py proj5 2>&1 | grep -v amdgpu.ids | tee "$OUT/proj_owner5_phases.txt"
timeout 600 python tools/ab_variants.py --op projection --variants=-1,-40 --cases proj,proj_fill,depth,depth_fill --flows smooth,iid 2>&1 | grep -v amdgpu.ids | tee "$OUT/proj_ab.txt"
timeout 300 python tools/probes/proj_burst.py 2>&1 | grep -v amdgpu.ids | tee "$OUT/proj_burst.txt"
echo "== RGB backward: without the image gradient (gradinput1 NULL) next to the whole backward, one process"
timeout 600 python tools/ab_variants.py --op fi_bwd --variants=-1 --cases fi_bwd_c2,fi_bwd_c2_nog1,fi_bwd,fi_bwd_nog1 --flows smooth 2>&1 | grep -v amdgpu.ids | tee "$OUT/fi_bwd_nog1.txt"
echo "== large motion: the projection and every operator against flow scale and camera pans"
timeout 300 python tools/probes/proj_motion_sweep.py 2>&1 | grep -v amdgpu.ids | tee "$OUT/proj_motion_sweep.txt"
timeout 300 python tools/probes/motion_sweep_all.py 2>&1 | grep -v amdgpu.ids | tee "$OUT/motion_sweep_all.txt"
echo "== race screen"; timeout 600 python tools/stress_projection.py 60 2>&1 | tail -3 | tee "$OUT/stress.log"
echo "== projection kernels of a call (kernel trace)"
( cd /tmp && export TMPDIR=/tmp && for kind in smooth iid; do
  timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/prof_proj_$kind" -o proj -- python "$REPO/tools/probes/proj_calls.py" $kind 60 2>&1 | grep "flow=" | tee -a "$OUT/proj_calls.txt"
  python "$REPO/tools/probes/proj_calls_summary.py" "$OUT/prof_proj_$kind/proj_results.db" 150 | tee -a "$OUT/proj_calls.txt"
  rm -rf "$OUT/prof_proj_$kind"; done )
echo "== SQ counters of the owner kernel"
bash tools/pmc_sq.sh $TAG/sq proj "proj_owner5<false" 2>&1 | tail -45 | tee "$OUT/proj_owner5_sq.txt"
cd /tmp && export TMPDIR=/tmp
echo "== rocprof kernel trace of bench.py (the same command as the bench line above, minus the CPU baseline and the secondary rows)"
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/prof_trace" -o bench -- python "$REPO/bench.py" --no-cpu-baseline --no-secondary > "$OUT/prof_trace.log" 2>&1
grep '^{"metric"' "$OUT/prof_trace.log" | tail -1 > "$OUT/bench_line_profiled_run.json"    # the bench line of THIS process
python "$REPO/tools/prof_summary.py" stats "$OUT/prof_trace/bench_results.db" --tail 300 --out "$OUT/bench_kernel_stats.txt" | grep -v "^at::\|^$" | head -6
rm -rf "$OUT/prof_trace"
echo "== rocprof kernel trace of the operator sweep"
timeout 900 rocprofv3 --kernel-trace --stats -d "$OUT/prof_sweep" -o sweep -- python "$REPO/tools/bench_ops.py" --quick --json "$OUT/bench_ops_profiled.json" > "$OUT/prof_sweep.log" 2>&1
python "$REPO/tools/prof_summary.py" stats "$OUT/prof_sweep/sweep_results.db" --out "$OUT/sweep_kernel_stats.txt" | grep -v "^at::\|^$" | head -40
rm -rf "$OUT/prof_sweep"
echo "== PMC traffic (FETCH_SIZE / WRITE_SIZE in separate passes, calibrated on a known copy)"
cd "$REPO" && timeout 900 python tools/pmc_traffic.py --out "$OUT" 2>&1 | tail -30
echo "== PMC traffic of the other kernels"
timeout 1500 python tools/pmc_traffic_ops.py --out "$OUT" 2>&1 | tail -14
ls "$OUT"
