#!/bin/bash
# lab build: the finish selection's per-phase clocks (block 0), the default bench shape, a fuzz slice and the filter tests
export FSGPU_BUILD_DEFS="-DFSGPU_EXPERIMENTS"
FSGPU_SELECT_STAMPS=1 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-two-tier --no-adversarial --no-encoders 2>/tmp/err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('qps=%.0f step=%.3fms main=%.4fms fb=%s' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['config'].get('exact_fallback_queries')))"
grep "select stamps" /tmp/err.txt | tail -3
python scripts/fuzz_batched.py ${1:-44} 60 | tail -1
python -m pytest tests/test_gpu_int8_filter.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -2
