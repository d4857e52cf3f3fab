#!/bin/bash
# the append-free sample's size on small slabs (FSGPU_RB_PCT, variant "expvi"): 1.25M and 2.5M rows
O=${1:-gpurun_out/r04rbs}; mkdir -p $O; export TMPDIR=/tmp
cp frankensearch_amd/libfsgpu.so /tmp/base.so
cp frankensearch_amd/libfsgpu_variant_expvi.so frankensearch_amd/libfsgpu.so
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  rows=%d qps=%.0f step=%.4fms main=%.4fms' % (d['config']['rows'], d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms']))"; }
{
for pct in 100 150 200 300 400 100 200; do
  echo "FSGPU_RB_PCT=$pct"
  for rows in 1250000 2500000; do FSGPU_RB_PCT=$pct python bench.py --rows $rows --steps 100 --warmup 10 --no-cpu-baseline --no-two-tier --no-adversarial --no-encoders 2>/dev/null | tail -1 | line; done
done
} 2>&1 | tee $O/rb_sweep_shard.txt
cp /tmp/base.so frankensearch_amd/libfsgpu.so
