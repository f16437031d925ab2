#!/bin/bash
# Round 6, GPU session 2: the pan + convergence test again; where a 20-step window of a 4-frame shard loses its 5 % (first launches
# after the synchronize, host enqueue, fixed cost); what the config-4 row's 65 s of set-up are (MIOpen compiles its kernels on
# a fresh box: the image has no .kdb) and whether a kernel cache directory carried in the tree removes them.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r06_s2
mkdir -p "$OUT"
cd "$REPO"
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests -q -m gpu -x -k "convergence or far_sources" 2>&1 | tail -3 | tee "$OUT/pytest.log"
for mode in eager graph; do for st in 20 300; do
  timeout 300 python bench.py --batch 4 --steps $st --warmup 5 --launch $mode --no-cpu-baseline --no-secondary 2>&1 | tail -1 | tee -a "$OUT/bench_b4.log" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']; r=d['roofline']
print('batch %2d steps %3d %-9s ms/step %.4f gpu_us/step %.2f enqueue_us %.1f fixed_us %.1f kernel_us %.2f first %s windows %s' % (c['batch_per_gpu'], d['steps'], c['launch'], d['ms_per_step'], c['window_gpu_us_per_step'], c['window_host_enqueue_us'], c['window_fixed_cost_us'], r['avg_launch_us'], r['first_launches_us'], c['window_ms_min_max']))"
done; done
echo "== config 4 set-up: cold, then with the kernel cache of the first run"
cat > /tmp/c4.py <<'PY'
import os, sys, time
t0 = time.perf_counter()
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "memc-net_amd"))
import torch
import my_package._ext.my_lib as my_lib
import networks
dev = torch.device("cuda:0")
torch.zeros(1, device=dev); torch.cuda.synchronize()
t1 = time.perf_counter()
torch.manual_seed(1)
with torch.device(dev):
    net = networks.MEMC_Net_star(channel=3, filter_size=4, training=False)
net = net.to(dev).eval()
torch.cuda.synchronize(); t2 = time.perf_counter()
frames = torch.rand((2, 4, 3, 720, 1280), device=dev)
with torch.no_grad():
    networks.interpolate_pairs(net, frames[0], frames[1]); torch.cuda.synchronize(); t3 = time.perf_counter()
    networks.interpolate_pairs(net, frames[0], frames[1]); torch.cuda.synchronize(); t4 = time.perf_counter()
print("import+context %.1f s, build net %.1f s, first pass %.1f s, second pass %.3f s" % (t1 - t0, t2 - t1, t3 - t2, t4 - t3))
PY
export MIOPEN_CUSTOM_CACHE_DIR=$OUT/miopen_cache MIOPEN_USER_DB_PATH=$OUT/miopen_cache
mkdir -p $MIOPEN_CUSTOM_CACHE_DIR
timeout 600 python /tmp/c4.py 2>&1 | grep -v amdgpu.ids | tee "$OUT/config4_setup.txt"
du -sh $MIOPEN_CUSTOM_CACHE_DIR; find $MIOPEN_CUSTOM_CACHE_DIR -type f | head -20
timeout 600 python /tmp/c4.py 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/config4_setup.txt"
ls -la ~/.cache/miopen 2>/dev/null; ls ~/.config/miopen 2>/dev/null
