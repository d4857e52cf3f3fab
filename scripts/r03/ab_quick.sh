#!/bin/bash
# default-shape bench, three times (same box): queries/s, step, main pass
for i in 1 2 3; do python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-two-tier --no-adversarial --no-encoders 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('qps=%.0f step=%.3fms main=%.4fms fb=%s' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['config'].get('exact_fallback_queries')))"; done
