#!/bin/bash
# The group-maxima sample stage (MfmaScanArgs::stage 3 + select_groups_kernel) against the two-stage thresholded sample it replaces:
# parity on the shipped build, then same-box A/B through the experiments build of vector_index.cpp (FSGPU_NO_GROUP_SAMPLE=1 = the old stages).
O=${1:-gpurun_out/r04gs}; mkdir -p $O; export TMPDIR=/tmp
( time python -m pytest tests/test_gpu_parity.py tests/test_gpu_int8_filter.py tests/test_gpu_fullsize.py -m gpu -q -x ) > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
python scripts/fuzz_batched.py 505 120 > $O/fuzz.txt 2>&1; tail -2 $O/fuzz.txt
cp frankensearch_amd/libfsgpu.so /tmp/base.so
cp frankensearch_amd/libfsgpu_variant_expvi.so frankensearch_amd/libfsgpu.so
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  qps=%.0f step=%.4fms main=%.4fms frac=%.3f fb=%s refiltered=%s' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['config'].get('exact_fallback_queries'), d['config'].get('filter_refiltered_on_f16_queries')))"; }
{
for off in "" 1 "" 1; do
  echo "old two-stage sample: ${off:-no}"
  env ${off:+FSGPU_NO_GROUP_SAMPLE=1} python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-two-tier --no-adversarial --no-encoders 2>/dev/null | tail -1 | line
  echo "  shard 1.25M:"; env ${off:+FSGPU_NO_GROUP_SAMPLE=1} python bench.py --rows 1250000 --steps 100 --warmup 10 --no-cpu-baseline --no-two-tier --no-adversarial --no-encoders 2>/dev/null | tail -1 | line
done
} 2>&1 | tee $O/ab.txt
cp /tmp/base.so frankensearch_amd/libfsgpu.so
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-two-tier --no-encoders 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('adversarial', {k: (round(v['queries_per_sec']), v['refiltered_on_f16_queries'], v['exact_fallback_queries'], v['batched_equals_oracle_rows_and_bits']) for k, v in d['adversarial_corpora'].items()})" | tee $O/adversarial.txt
python scripts/r04/filtered_tput.py 2>&1 | tail -5 | tee $O/filtered.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-two-tier --no-adversarial --no-encoders > $O/bench_traced.json 2> $O/trace.err
head -10 $O/trace/*kernel_stats.csv | cut -c1-170 | tee $O/trace_head.txt
