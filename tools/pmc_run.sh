#!/bin/bash
# usage: pmc_run.sh <tag> <variants> ; runs FETCH_SIZE, WRITE_SIZE, TCC hit/miss passes over bench_ops quick
TAG=$1; VARS=$2
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for CTR in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  N=$(echo $CTR | tr ' ' '_')
  timeout 300 rocprofv3 --pmc $CTR --kernel-trace -d $OUT/pmc_$N -o r -- python $REPO/tools/bench_ops.py --only copy,fi_fwd --quick --headline-only --variants $VARS --json $OUT/ops_$N.json > $OUT/pmc_$N.log 2>&1
  python $REPO/tools/prof_summary.py pmc $OUT/pmc_$N/r_results.db --out $OUT/pmc_$N.json > /dev/null
  rm -rf $OUT/pmc_$N
done
python - <<PY
import json,glob
for f in sorted(glob.glob('$OUT/pmc_*.json')):
    for r in json.load(open(f)):
        if ('fi_fwd' in r['kernel'] or 'copyBuffer' in r['kernel']) and r['grid'] > 2e6:
            print('%-40s %-22s n=%-3d grid=%-9d mean=%-10.4g dur_us=%.1f' % (r['kernel'][:40], r['counter'], r['dispatches'], r['grid'], r['mean_value'], r['mean_duration_us']))
PY
