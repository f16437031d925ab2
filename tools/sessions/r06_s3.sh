#!/bin/bash
# Round 6, GPU session 3: the three kernels that no longer spill (tests + A/B against round 5's library in one process), the
# bench line's windows without events inside the clock (eager at every shard size), the config-4 row on the shipped MIOpen cache.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r06_s3
mkdir -p "$OUT"
cd "$REPO"
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests -q -m gpu -x -k "many_channel or ctx or convergence or bench_line or context or abi or capture" 2>&1 | tail -4 | tee "$OUT/pytest.log"
echo "== A/B: working tree (A) against round 5's last commit (B)"
timeout 900 python tools/ab_libs.py memc-net_amd/lib/libmemc_hip.so tools/probes/variants/libmemc_hip_round5.so --op fi_bwd_c64,interp_bwd_c64,ctx_img_blend,fi_fwd --rounds 5 --iters 8 2>&1 | grep -v amdgpu.ids | tee "$OUT/spills_ab.txt"
echo "== shard sizes, steps 20 vs 300"
for b in 32 16 8 4; do for st in 20 300; do
  timeout 300 python bench.py --batch $b --steps $st --warmup 5 --no-cpu-baseline --no-secondary 2>&1 | tail -1 | tee -a "$OUT/bench_shards_steps.log" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']; r=d['roofline']
print('batch %2d steps %3d %-9s value %9.1f ms/step %.4f gpu_us/step %.2f enqueue_us %.1f fixed_us %.1f kernel_us %.2f first %s windows %s' % (c['batch_per_gpu'], d['steps'], c['launch'], d['value'], d['ms_per_step'], c['window_gpu_us_per_step'], c['window_host_enqueue_us'], c['window_fixed_cost_us'], r['avg_launch_us'], r['first_launches_us'], c['window_ms_min_max']))"
done; done
echo "== 2 ranks sharing the GPU (gloo), steps 20 / 300"
for st in 20 300; do
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29641 bench.py --gpus 2 --share-gpu --batch 8 --steps $st --warmup 5 2>&1 | grep '^{"metric"' | tee -a "$OUT/bench_share_gpu.log" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']; r=d['roofline']
print('2 ranks on one GPU: steps %3d ms/step %.4f barrier_us %s per-rank %s' % (d['steps'], d['ms_per_step'], d['dist']['barrier_us'], d['dist']['per_rank_ms_per_step']))"
done
echo "== the default bench line (time it)"
/usr/bin/time -v timeout 900 python bench.py > "$OUT/bench.log" 2> "$OUT/bench.err"; grep "Elapsed" "$OUT/bench.err"; tail -1 "$OUT/bench.log" | cut -c1-500
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06_s3/bench.log").read().strip().splitlines()[-1])
print(json.dumps(d["secondary"].get("config4_memc_net_star_4x1280x720"), indent=1))
PY
echo "== driver's arguments"
/usr/bin/time -v timeout 900 python bench.py --steps 20 --warmup 5 > "$OUT/bench_driver_args.log" 2> "$OUT/bench_driver_args.err"; grep "Elapsed" "$OUT/bench_driver_args.err"; tail -1 "$OUT/bench_driver_args.log" | cut -c1-900
