#!/bin/bash
# GPU session 3 of round 4
O=gpurun_out/r04s3; mkdir -p $O
( time python -m pytest tests -m gpu -q -x ) > $O/pytest_all.txt 2>&1; tail -6 $O/pytest_all.txt
python scripts/fuzz_batched.py 92 80 > $O/fuzz_batched.txt 2>&1; tail -1 $O/fuzz_batched.txt
cp frankensearch_amd/libfsgpu.so /tmp/base.so
python scripts/r04/bitmap_soak.py 240 300 > $O/soak_default.txt 2>&1; echo "== default"; tail -2 $O/soak_default.txt
for v in bmcheck noinv; do
  cp frankensearch_amd/libfsgpu_variant_$v.so frankensearch_amd/libfsgpu.so
  python scripts/r04/bitmap_soak.py 420 100 > $O/soak_$v.txt 2>&1; echo "== $v"; tail -25 $O/soak_$v.txt
done
run() { for i in 1 2; do python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-two-tier --no-adversarial --no-encoders $1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  qps=%.0f step=%.3fms main=%.4fms fb=%s' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['config'].get('exact_fallback_queries')))"; done; }
for v in base split base split; do
  if [ $v = base ]; then cp /tmp/base.so frankensearch_amd/libfsgpu.so; else cp frankensearch_amd/libfsgpu_variant_$v.so frankensearch_amd/libfsgpu.so; fi
  echo "== $v 10M"; run ""; echo "== $v 1.25M"; run "--rows 1250000"
done 2>&1 | tee $O/ab_split.txt
cp /tmp/base.so frankensearch_amd/libfsgpu.so
python scripts/r04/outlier_census.py 8 > $O/outlier_census.txt 2>&1; tail -40 $O/outlier_census.txt
