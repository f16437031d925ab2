mkdir -p gpurun_out/r06_s29; for b in 16 32; do for s in 1 2 3; do python bench.py --batch $b --input-sets $s --steps 100 --warmup 20 --no-cpu-baseline --no-secondary 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('batch', d['config']['batch_per_gpu'], 'input_sets', d['config']['input_sets'], 'value', d['value'], 'kernel_us', d['roofline']['avg_launch_us'], 'frac', d['roofline']['frac'])" | tee -a gpurun_out/r06_s29/input_sets.txt; done; done
