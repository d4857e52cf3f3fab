#!/bin/bash
# same-box A/B of library variants on the single-query latency at 1M / 1.25M rows and the exact kernel at 10M
cd "$(dirname "$0")/../.."
cp frankensearch_amd/libfsgpu.so /tmp/base.so
run() { for rows in 1000000 1250000; do python bench.py --rows $rows --steps 20 --warmup 5 --no-cpu-baseline --no-two-tier --no-adversarial --no-encoders 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('  rows', $rows, 'p50 single ms', round(d['p50_latency_ms_single_query'],4))"; done
python bench.py --exact --batch 1 --steps 60 --warmup 5 --no-cpu-baseline --no-two-tier 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('  10M exact launch ms', round(d['roofline']['avg_launch_ms'],4), 'frac', round(d['roofline']['frac'],4))"; }
echo base; run
for v in frankensearch_amd/libfsgpu_variant*.so; do cp $v frankensearch_amd/libfsgpu.so; echo $v; run; done
cp /tmp/base.so frankensearch_amd/libfsgpu.so; echo base again; run
