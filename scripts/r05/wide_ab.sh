#!/bin/bash
# Same-box A/B of main-pass builds (round 5): the shipped libfsgpu.so against frankensearch_amd/libfsgpu_variant_NAME.so
# (scripts/r04/build_variant.sh; "r04" = the library as round 4 shipped it).  Parity first (suite subset + batched fuzzer on the shipped
# build), then the bench shape and the 1.25M-row shard per build, then s_memtime stamps of the shipped main pass (variant "exp").
#   scripts/r05/wide_ab.sh OUTDIR NAME...
O=${1:-gpurun_out/r05ab}; shift; mkdir -p $O; export TMPDIR=/tmp
( time python -m pytest tests/test_gpu_parity.py tests/test_gpu_int8_filter.py -m gpu -q -x ) > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
python scripts/fuzz_batched.py 505 120 > $O/fuzz.txt 2>&1; tail -2 $O/fuzz.txt
cp frankensearch_amd/libfsgpu.so /tmp/base.so
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('  qps=%.0f step=%.4fms main=%.4fms frac=%.3f fb=%s' % (d['value'], d['ms_per_step'], r['avg_launch_ms'], r['frac'], d['config'].get('exact_fallback_queries')))"; }
run() {
  for i in 1 2; do python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-two-tier --no-adversarial --no-encoders 2>/dev/null | tail -1 | line; done
  echo "  shard 1.25M:"; for i in 1 2; do python bench.py --rows 1250000 --steps 60 --warmup 10 --no-cpu-baseline --no-two-tier --no-adversarial --no-encoders 2>/dev/null | tail -1 | line; done
}
{
echo "shipped"; run
for v in "$@"; do case $v in exp*) continue;; esac; cp frankensearch_amd/libfsgpu_variant_$v.so frankensearch_amd/libfsgpu.so; echo "variant $v"; run; done
cp /tmp/base.so frankensearch_amd/libfsgpu.so; echo "shipped again"; run
} 2>&1 | tee $O/ab.txt
for v in "$@"; do
  case $v in exp*) ;; *) continue;; esac
  cp frankensearch_amd/libfsgpu_variant_$v.so frankensearch_amd/libfsgpu.so
  for rows in 10000000 1250000; do
    echo "== variant $v: $rows rows x 384, 1,024 queries per launch (two 512-query groups)"
    FSGPU_WIDE_DBG=8 python bench.py --rows $rows --steps 8 --warmup 3 --blocking-steps --no-cpu-baseline --no-two-tier --no-adversarial --no-encoders 2>&1 >/dev/null | grep "wide stamps" | tail -3
  done
  cp /tmp/base.so frankensearch_amd/libfsgpu.so
done | tee $O/wide_stamps.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-two-tier --no-adversarial --no-encoders > $O/bench_traced.json 2> $O/trace.err
head -12 $O/trace/*kernel_stats.csv | cut -c1-180 | tee $O/trace_head.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-two-tier --no-encoders 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('adversarial', json.dumps(d.get('adversarial_corpora'))[:1500])" | tee $O/adversarial.txt
