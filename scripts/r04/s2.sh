#!/bin/bash
# GPU session 2 of round 4: new two-tier / sharded tests, diagnostic soaks of the bitmap loads, flags A/B, adversarial corpora, gloo rehearsal
O=gpurun_out/r04s2; mkdir -p $O
( time python -m pytest tests/test_gpu_two_tier.py tests/test_gpu_sharded.py tests/test_gpu_int8_filter.py tests/test_gpu_config3_fullsize.py -m gpu -x -q ) > $O/pytest_new.txt 2>&1; tail -15 $O/pytest_new.txt
python scripts/fuzz_batched.py 91 60 > $O/fuzz_batched.txt 2>&1; tail -2 $O/fuzz_batched.txt
cp frankensearch_amd/libfsgpu.so /tmp/base.so
cp frankensearch_amd/libfsgpu_variant_bmcheck.so frankensearch_amd/libfsgpu.so
python scripts/r04/bitmap_soak.py 480 100 > $O/soak_bmcheck.txt 2>&1; echo "== bmcheck"; tail -40 $O/soak_bmcheck.txt
cp frankensearch_amd/libfsgpu_variant_noglc.so frankensearch_amd/libfsgpu.so
python scripts/r04/bitmap_soak.py 300 100 > $O/soak_noglc.txt 2>&1; echo "== noglc"; tail -12 $O/soak_noglc.txt
run() { for i in 1 2; do python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-two-tier --no-adversarial --no-encoders $1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  qps=%.0f step=%.3fms main=%.4fms fb=%s' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['config'].get('exact_fallback_queries')))"; done; }
for v in base flags base; do
  if [ $v = base ]; then cp /tmp/base.so frankensearch_amd/libfsgpu.so; else cp frankensearch_amd/libfsgpu_variant_$v.so frankensearch_amd/libfsgpu.so; fi
  echo "== $v 10M"; run ""; echo "== $v 1.25M"; run "--rows 1250000"
done 2>&1 | tee $O/ab_flags.txt
cp /tmp/base.so frankensearch_amd/libfsgpu.so
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-two-tier --no-encoders 2>/dev/null | tail -1 > $O/bench_adv.json
python - <<PY
import json
d = json.loads(open("$O/bench_adv.json").read().strip().splitlines()[-1])
for kname, a in d["adversarial_corpora"].items():
    print(kname, {x: a[x] for x in ("queries_per_sec", "int8_filter_active_after", "int8_filter_queries", "refiltered_on_f16_queries", "exact_fallback_queries", "batched_equals_oracle_rows_and_bits", "batched_equals_exact_kernels_64_queries")})
PY
( time FSGPU_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 10 --warmup 3 --config5-rows 6000000 ) > $O/rehearsal_gloo2.txt 2>&1; tail -c 3000 $O/rehearsal_gloo2.txt
( time python -m pytest tests -m gpu -q -x ) > $O/pytest_all.txt 2>&1; tail -8 $O/pytest_all.txt
