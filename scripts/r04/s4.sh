#!/bin/bash
# GPU session 4 of round 4: expect-hint A/B, new tests, filtered throughput, outlier corpus, one more no-invalidate soak
O=gpurun_out/r04s4; mkdir -p $O
cp frankensearch_amd/libfsgpu.so /tmp/base.so
run() { for i in 1 2; do python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-two-tier --no-adversarial --no-encoders $1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  qps=%.0f step=%.3fms main=%.4fms fb=%s' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['config'].get('exact_fallback_queries')))"; done; }
for v in base noexpect base noexpect; do
  if [ $v = base ]; then cp /tmp/base.so frankensearch_amd/libfsgpu.so; else cp frankensearch_amd/libfsgpu_variant_$v.so frankensearch_amd/libfsgpu.so; fi
  echo "== $v 10M"; run ""; echo "== $v 1.25M"; run "--rows 1250000"
done 2>&1 | tee $O/ab_expect.txt
cp /tmp/base.so frankensearch_amd/libfsgpu.so
( time python -m pytest tests -m gpu -q -x ) > $O/pytest_all.txt 2>&1; tail -6 $O/pytest_all.txt
python scripts/fuzz_batched.py 93 60 > $O/fuzz_batched.txt 2>&1; tail -1 $O/fuzz_batched.txt
python scripts/r04/filtered_tput.py > $O/filtered_tput.txt 2>&1; tail -6 $O/filtered_tput.txt
python scripts/r04/outlier_census.py 8 > $O/outlier_census.txt 2>&1; grep "^step" $O/outlier_census.txt
cp frankensearch_amd/libfsgpu_variant_noinv.so frankensearch_amd/libfsgpu.so
python scripts/r04/bitmap_soak.py 420 100 > $O/soak_noinv.txt 2>&1; echo "== noinv"; tail -12 $O/soak_noinv.txt
cp /tmp/base.so frankensearch_amd/libfsgpu.so
