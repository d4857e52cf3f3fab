#!/bin/bash
# What a step pays around its kernels: event pairs around every main launch against every 4th, blocking wait against the polled wait
# (fsgpu_index_set_spin_wait) — bench shape and a 1.25M-row shard, same box.   scripts/r04/step_overheads.sh OUTDIR
O=${1:-gpurun_out/r04step}; mkdir -p $O; export TMPDIR=/tmp
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  qps=%.0f step=%.4fms main=%.4fms frac=%.3f launches=%s' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['roofline']['launches']))"; }
{
for spin in 0 5000 0 5000; do
  echo "spin_wait_us=$spin (events around every 4th main launch)"
  python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-two-tier --no-adversarial --no-encoders 2>/dev/null | tail -1 | line
  echo "  shard 1.25M:"; python bench.py --rows 1250000 --steps 100 --warmup 10 --no-cpu-baseline --no-two-tier --no-adversarial --no-encoders 2>/dev/null | tail -1 | line
done
echo "spin_wait_us=5000, 12 steps (events around every main launch)"
python bench.py --steps 12 --warmup 10 --no-cpu-baseline --no-two-tier --no-adversarial --no-encoders 2>/dev/null | tail -1 | line
echo "  shard 1.25M:"; python bench.py --rows 1250000 --steps 12 --warmup 10 --no-cpu-baseline --no-two-tier --no-adversarial --no-encoders 2>/dev/null | tail -1 | line
} 2>&1 | tee $O/step_overheads.txt
python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -2
