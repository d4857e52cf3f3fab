#!/bin/bash
# same-box A/B of prebuilt library variants: frankensearch_amd/libfsgpu_variant*.so against libfsgpu.so (each: 2 bench runs)
cd "$(dirname "$0")/../.."
cp frankensearch_amd/libfsgpu.so /tmp/base.so
run() { for i in 1 2; do python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-two-tier --no-adversarial --no-encoders 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  qps=%.0f step=%.3fms main=%.4fms fb=%s' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['config'].get('exact_fallback_queries')))"; done; }
echo base; run
for v in frankensearch_amd/libfsgpu_variant*.so; do cp $v frankensearch_amd/libfsgpu.so; echo $v; run; done
cp /tmp/base.so frankensearch_amd/libfsgpu.so; echo base again; run
