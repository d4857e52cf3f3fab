#!/bin/bash
# The sample's size against the main pass's append traffic, again after round 5's cheaper append path (r04: profiles/r04/rb_sweep*.txt):
# FSGPU_RB_PCT (experiments build of the vector_index translation units, variant "expvi") scales the sample the plan chose; one box,
# the bench shape, a 1.25M-row shard and a 2.5M-row shard.   scripts/r05/rb_sweep.sh OUTDIR
O=${1:-gpurun_out/r05rb}; mkdir -p $O; export TMPDIR=/tmp
cp frankensearch_amd/libfsgpu.so /tmp/base.so
cp frankensearch_amd/libfsgpu_variant_expvi.so frankensearch_amd/libfsgpu.so
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  rows=%d qps=%.0f step=%.4fms main=%.4fms frac=%.3f fb=%s' % (d['config']['rows'], d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['config'].get('exact_fallback_queries')))"; }
{
for pct in ${PCTS:-100 50 65 80 125 100}; do
  echo "FSGPU_RB_PCT=$pct"
  for rows in 10000000 2500000 1250000; do
    FSGPU_RB_PCT=$pct python bench.py --rows $rows --steps 60 --warmup 10 --no-cpu-baseline --no-two-tier --no-adversarial --no-encoders 2>/dev/null | tail -1 | line
  done
done
} 2>&1 | tee $O/rb_sweep.txt
cp /tmp/base.so frankensearch_amd/libfsgpu.so
