#!/bin/bash
# A/B of libfsgpu variants on one box: default vs frankensearch_amd/libfsgpu_variant_$1.so — 10M step, 1.25M-row shard step, the two-tier stages
L=frankensearch_amd/libfsgpu.so
cp $L /tmp/libfsgpu_default.so
for v in default "$@" default "$@"; do
  [ $v = default ] && cp /tmp/libfsgpu_default.so $L || cp frankensearch_amd/libfsgpu_variant_$v.so $L
  for rows in 10000000 1250000; do
    python bench.py --rows $rows --steps 60 --warmup 10 --no-cpu-baseline --no-two-tier --no-adversarial --no-encoders 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$v rows=$rows', 'ms_per_step %.4f' % d['ms_per_step'], 'main %.4f' % d['roofline']['avg_launch_ms'])"
  done
  python scripts/r06/prof_two_tier_stages.py 2>/dev/null | grep "ms per" | sed "s/^/$v  /"
done
cp /tmp/libfsgpu_default.so $L
