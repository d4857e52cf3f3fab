#!/bin/bash
# Same-box A/B of main-pass builds: the shipped libfsgpu.so against frankensearch_amd/libfsgpu_variant_NAME.so (scripts/r04/build_variant.sh)
# — parity first (suite subset + fuzzer on the shipped build), then the bench shape and the 1.25M-row shard per build, the adversarial
# corpora, and the counters of the shipped main pass beside its skeletons (variant "exp").   scripts/r04/wide_ab.sh OUTDIR NAME...
O=${1:-gpurun_out/r04ab}; shift; mkdir -p $O; export TMPDIR=/tmp
( time python -m pytest tests/test_gpu_parity.py tests/test_gpu_int8_filter.py -m gpu -q -x ) > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
python scripts/fuzz_batched.py 404 90 > $O/fuzz.txt 2>&1; tail -2 $O/fuzz.txt
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
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-two-tier --no-adversarial --no-encoders > $O/bench_traced.json 2> $O/trace.err
head -12 $O/trace/*kernel_stats.csv | cut -c1-180 | tee $O/trace_head.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-two-tier --no-encoders 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('adversarial', json.dumps(d.get('adversarial_corpora'))[:1500])" | tee $O/adversarial.txt
[ -f frankensearch_amd/libfsgpu_variant_exp.so ] && bash scripts/r04/skeleton_clocks.sh $O/skc 0 16 31 2>&1 | tail -4
