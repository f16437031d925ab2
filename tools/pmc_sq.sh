#!/bin/bash
# tools/pmc_sq.sh <tag> <bench_ops --only key> <kernel-name substring> -- shader-core counters of one operator
# (rocprofv3 --pmc in two passes of <= 8 SQ counters; only --kernel-trace beside them).  Run on the GPU box.
# Prints per counter the mean per dispatch of the LARGEST grid and a few derived ratios.
TAG=$1; ONLY=$2; MATCH=$3; EXTRA=${4:-}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P1="SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
P2="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_LDS_ATOMIC SQ_INSTS_VMEM SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_WAIT_ANY SQ_WAIT_INST_ANY"
# round 4: the SCALAR pipe (its issue rate per SIMD equals the vector pipe's) and the wave count the per-wave figures need
P3="SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_WAVES SQ_INSTS_VALU"
i=0
for CTRS in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  # (PMC_CMD: another workload than bench_ops.py, e.g. "python tools/probes/proj_far_load.py 4")
  timeout 300 rocprofv3 --pmc $CTRS --kernel-trace -d $OUT/p$i -o r -- ${PMC_CMD:-python $REPO/tools/bench_ops.py --only $ONLY $EXTRA --json $OUT/ops$i.json} > $OUT/p$i.log 2>&1
  python $REPO/tools/prof_summary.py pmc $OUT/p$i/r_results.db --match "$MATCH" --out $OUT/pmc$i.json > /dev/null
  rm -rf $OUT/p$i
done
python - <<PY
import json
rows = json.load(open('$OUT/pmc1.json')) + json.load(open('$OUT/pmc2.json')) + json.load(open('$OUT/pmc3.json'))
g = max(r['grid'] for r in rows)
v = {r['counter']: r['mean_value'] for r in rows if r['grid'] == g}
k = [r for r in rows if r['grid'] == g][0]
print(k['kernel'][:60], 'grid', int(g), 'vgpr', k['vgpr'], 'lds', k['lds_bytes'], 'scratch', k['scratch'], 'dur_us %.1f' % k['mean_duration_us'])
for c in sorted(v): print('  %-26s %.4g' % (c, v[c]))
busy = v.get('SQ_BUSY_CU_CYCLES')
if busy:
    for c in ('SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_SCA', 'SQ_ACTIVE_INST_LDS', 'SQ_ACTIVE_INST_VMEM', 'SQ_LDS_BANK_CONFLICT', 'SQ_LDS_IDX_ACTIVE', 'SQ_WAIT_INST_LDS'):
        if c in v: print('  %-26s / SQ_BUSY_CU_CYCLES = %.3f' % (c, v[c] / busy))
waves = v.get('SQ_WAVES')
if waves:
    for c in ('SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_INSTS_SMEM', 'SQ_INSTS_BRANCH', 'SQ_INSTS_LDS', 'SQ_INSTS_LDS_ATOMIC', 'SQ_INSTS_VMEM'):
        if c in v: print('  %-26s / wave = %.1f' % (c, v[c] / waves))
json.dump({'kernel': k['kernel'], 'grid': g, 'counters': v}, open('$OUT/sq_summary.json', 'w'), indent=1)
PY
