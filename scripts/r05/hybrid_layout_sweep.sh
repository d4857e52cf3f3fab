#!/bin/bash
# Which query-group x row-shard layout answers 1,024 queries over 10M x 384 fastest on N GPUs?  One GPU rehearses ONE rank's share of
# every layout (its rows = 10M / row shards, its queries = 1,024 / groups per step; begin / end loop, no exchange): the world's step is
# that rank's step (+ the exchange, ~0.03 ms hidden under the next scan).   scripts/r05/hybrid_layout_sweep.sh OUTDIR
O=${1:-gpurun_out/r05hyb}; mkdir -p $O; export TMPDIR=/tmp
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('step=%.4fms main=%.4fms frac=%.3f fb=%s' % (d['ms_per_step'], r['avg_launch_ms'], r['frac'], d['config'].get('exact_fallback_queries')))"; }
{
for n in 2 4 8; do
  for g in 1 2 4 8; do
    [ $g -gt $n ] && continue
    s=$((n / g)); rows=$((10000000 / s)); b=$((1024 / g))
    for rep in 1 2; do
      printf "N=%d  %d groups x %d row shards: per rank %8d rows x %4d queries  " $n $g $s $rows $b
      python bench.py --rows $rows --batch $b --steps 60 --warmup 10 --no-cpu-baseline --no-two-tier --no-adversarial --no-encoders 2>/dev/null | tail -1 | line
    done
  done
done
} 2>&1 | tee $O/hybrid_layout_sweep.txt
