#!/bin/bash
# kernel trace of the single-query latency path at 1M rows: what the 0.18 ms consist of
OUT=gpurun_out/lat1m; mkdir -p $OUT; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python bench.py --rows 1000000 --exact --batch 1 --steps 50 --warmup 5 --no-cpu-baseline --no-two-tier --no-adversarial --no-encoders > $OUT/trace.json 2> $OUT/trace.err
python - <<PY
import csv, glob
for f in glob.glob("$OUT/trace/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    for r in rows[:8]:
        print(f'{r["Name"][:80]:80s} calls={r["Calls"]:>5s} avg_us={float(r["AverageNs"])/1e3:9.1f}')
PY
tail -1 $OUT/trace.json | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('p50 single ms (under trace)', d.get('p50_latency_ms_single_query'), 'ms/step', d['ms_per_step'], d['roofline'].get('avg_launch_ms'))"
