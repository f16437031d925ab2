#!/bin/bash
# Round 6, GPU session 9: the collective layer on a one-rank RCCL communicator (--dist-single) and the bench-line tests.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r06_s9
mkdir -p "$OUT"
cd "$REPO"
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python bench.py --dist-single --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>&1 | tail -3 | tee "$OUT/bench_dist_single.log" | cut -c1-1800
timeout 900 python -m pytest tests/test_gpu_bench_line.py -q -x 2>&1 | tail -5 | tee "$OUT/pytest.log"
T0=$(date +%s.%N)
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29641 bench.py --gpus 2 --share-gpu --batch 8 --steps 20 --warmup 5 2>&1 | grep '^{"metric"' | cut -c1-200
python -c "import sys; print('2-rank gloo run end to end: %.1f s' % (float(sys.argv[2]) - float(sys.argv[1])))" $T0 $(date +%s.%N) | tee "$OUT/two_rank_wall.txt"
