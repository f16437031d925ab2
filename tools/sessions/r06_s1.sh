#!/bin/bash
# Round 6, GPU session 1: (a) the new pan + convergence parity test and the projection tests on the packed plane that
# carries the motion residual; (b) bench.py's windowed timing: shard sizes at the driver's --steps 20 --warmup 5 against
# --steps 300, eager and graph, and the 2-rank --share-gpu plumbing run at 20 steps; (c) the whole default bench line with the
# config-4 row; (d) the projection A/B: working tree against round 5's last commit, one process.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r06_s1
mkdir -p "$OUT"
cd "$REPO"
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== pytest (projection, bench line)"
timeout 1500 python -m pytest tests -q -m gpu -x -k "projection or bench_line or hole or pan" 2>&1 | tail -6 | tee "$OUT/pytest.log"
echo "== shard sizes, steps 20 vs 300"
for b in 32 16 8 4; do for st in 20 300; do
  timeout 300 python bench.py --batch $b --steps $st --warmup 5 --no-cpu-baseline --no-secondary 2>&1 | tail -1 | tee -a "$OUT/bench_shards_steps.log" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']; r=d['roofline']
print('batch %2d steps %3d %-9s value %9.1f ms/step %.4f gpu_us/step %.2f kernel_us %.2f windows %s' % (c['batch_per_gpu'], d['steps'], c['launch'], d['value'], d['ms_per_step'], c['window_gpu_us_per_step'], r['avg_launch_us'], c['window_ms_min_max']))"
done; done
echo "== batch 4, eager instead of graph"
for st in 20 300; do timeout 300 python bench.py --batch 4 --steps $st --warmup 5 --launch eager --no-cpu-baseline --no-secondary 2>&1 | tail -1 | tee -a "$OUT/bench_shards_steps.log" | cut -c1-330; done
echo "== 2 ranks sharing the GPU (gloo), steps 20 / 300"
for st in 20 300; do
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29641 bench.py --gpus 2 --share-gpu --batch 8 --steps $st --warmup 5 2>&1 | grep '^{"metric"' | tee -a "$OUT/bench_share_gpu.log" | cut -c1-420
done
echo "== the default bench line"
timeout 900 python bench.py 2>&1 | tail -1 > "$OUT/bench.log"; cut -c1-600 "$OUT/bench.log"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06_s1/bench.log").read())
print(json.dumps(d["secondary"].get("config4_memc_net_star_4x1280x720"), indent=1))
for k, v in d["secondary"].items():
    if "projection" in k: print(k, v)
PY
echo "== driver's arguments"
timeout 900 python bench.py --steps 20 --warmup 5 --no-secondary --cpu-seconds 2 2>&1 | tail -1 | cut -c1-700
echo "== projection A/B against round 5"
for extra in "" "--pan 40" "--scale 2"; do
timeout 600 python tools/ab_libs.py memc-net_amd/lib/libmemc_hip.so tools/probes/variants/libmemc_hip_round5.so $extra 2>&1 | grep -v amdgpu.ids | sed "s/^/[$extra] /" | tee -a "$OUT/proj_ab_round5.txt"
done
