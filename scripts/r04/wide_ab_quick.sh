#!/bin/bash
# scripts/r04/wide_ab_quick.sh OUTDIR NAME...: the bench shape and the 1.25M-row shard, shipped build against variants, no parity pass
O=${1:-gpurun_out/r04abq}; shift; mkdir -p $O; export TMPDIR=/tmp
cp frankensearch_amd/libfsgpu.so /tmp/base.so
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  qps=%.0f step=%.4fms main=%.4fms frac=%.3f fb=%s' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['config'].get('exact_fallback_queries')))"; }
run() {
  for i in 1 2; do python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-two-tier --no-adversarial --no-encoders 2>/dev/null | tail -1 | line; done
  echo "  shard 1.25M:"; for i in 1 2; do python bench.py --rows 1250000 --steps 60 --warmup 10 --no-cpu-baseline --no-two-tier --no-adversarial --no-encoders 2>/dev/null | tail -1 | line; done
}
{
echo "shipped"; run
for v in "$@"; do cp frankensearch_amd/libfsgpu_variant_$v.so frankensearch_amd/libfsgpu.so; echo "variant $v"; run; done
cp /tmp/base.so frankensearch_amd/libfsgpu.so; echo "shipped again"; run
} 2>&1 | tee $O/ab.txt
