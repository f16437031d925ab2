#!/bin/bash
# Round 6, GPU session 30: the ranks' shards on one GPU with the new input-set rule (every shard cycles 2.8 GB), at the driver's
# 20 steps and at 300; the two-rank gloo run on one GPU.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r06_s30
mkdir -p "$OUT"
cd "$REPO"
export HSA_ENABLE_IPC_MODE_LEGACY=0
for b in 32 16 8 4; do for st in 20 300; do timeout 300 python bench.py --batch $b --steps $st --warmup 5 --no-cpu-baseline --no-secondary 2>&1 | tail -1 | tee -a "$OUT/bench_shards.log" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']; r=d['roofline']
print('batch %2d steps %3d %-9s sets %d value %9.1f ms/step %.4f gpu_us/step %.2f enqueue_us %.1f fixed_us %.1f kernel_us %.2f frac %.3f first %s windows %s' % (c['batch_per_gpu'], d['steps'], c['launch'], c['input_sets'], d['value'], d['ms_per_step'], c['window_gpu_us_per_step'], c['window_host_enqueue_us'], c['window_fixed_cost_us'], r['avg_launch_us'], r['frac'], r['first_launches_us'], c['window_ms_min_max']))" | tee -a "$OUT/bench_shards_summary.txt"; done; done
for st in 20 300; do timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29641 bench.py --gpus 2 --share-gpu --batch 8 --steps $st --warmup 5 2>&1 | grep '^{"metric"' | tee -a "$OUT/bench_share_gpu.log" | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('2 ranks on one GPU: steps %3d sets %d ms/step %.4f barrier_us %s per-rank ms/step %s' % (d['steps'], d['config']['input_sets'], d['ms_per_step'], d['dist']['barrier_us'], d['dist']['per_rank_ms_per_step']))" | tee -a "$OUT/bench_shards_summary.txt"; done
echo "== bench at the driver's arguments, whole line (traffic record re-taken in r06c)"
timeout 900 python bench.py --steps 20 --warmup 5 2>&1 | tail -1 > "$OUT/bench_driver_args.json"; cut -c1-300 "$OUT/bench_driver_args.json"
timeout 900 python bench.py 2>&1 | tail -1 > "$OUT/bench_line.json"; cut -c1-300 "$OUT/bench_line.json"
