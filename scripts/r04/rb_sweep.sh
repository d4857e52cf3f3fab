#!/bin/bash
# The second sample's size against the main pass's append traffic: FSGPU_RB_PCT (experiments build of vector_index.cpp, variant "expvi")
# scales the sample the plan chose; one box, the bench shape and a 1.25M-row shard.   scripts/r04/rb_sweep.sh OUTDIR
O=${1:-gpurun_out/r04rb}; mkdir -p $O; export TMPDIR=/tmp
cp frankensearch_amd/libfsgpu.so /tmp/base.so
cp frankensearch_amd/libfsgpu_variant_expvi.so frankensearch_amd/libfsgpu.so
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  qps=%.0f step=%.4fms main=%.4fms frac=%.3f fb=%s' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['config'].get('exact_fallback_queries')))"; }
{
for pct in ${PCTS:-100 150 200 300 400 100}; do
  echo "FSGPU_RB_PCT=$pct"
  FSGPU_RB_PCT=$pct python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-two-tier --no-adversarial --no-encoders 2>/dev/null | tail -1 | line
  echo "  shard 1.25M:"; FSGPU_RB_PCT=$pct python bench.py --rows 1250000 --steps 60 --warmup 10 --no-cpu-baseline --no-two-tier --no-adversarial --no-encoders 2>/dev/null | tail -1 | line
done
} 2>&1 | tee $O/rb_sweep.txt
cp /tmp/base.so frankensearch_amd/libfsgpu.so
