#!/bin/bash
# single-query latency through the host-pointer ABI at the small sizes, + the exact-path parity tests
python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
for rows in 1000000 1250000 10000000; do
python bench.py --rows $rows --steps 20 --warmup 5 --no-cpu-baseline --no-two-tier --no-adversarial --no-encoders 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('rows', $rows, 'p50 single ms', round(d['p50_latency_ms_single_query'],4), 'qps', round(d['value']))"
done
bash scripts/r03/lat_trace.sh 2>&1 | grep -E "scan_topk|merge_topk|p50"
