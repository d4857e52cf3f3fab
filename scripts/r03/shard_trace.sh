#!/bin/bash
# kernel trace of a 1.25M-row shard's step (the 8-GPU operating point): what the 0.66 ms consist of
OUT=gpurun_out/shard1m25; mkdir -p $OUT; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python bench.py --rows 1250000 --steps 50 --warmup 5 --no-cpu-baseline --no-two-tier --no-adversarial --no-encoders > $OUT/trace.json 2> $OUT/trace.err
python - <<PY
import csv, glob, json
for f in glob.glob("$OUT/trace/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    for r in rows[:14]:
        print(f'{r["Name"][:84]:84s} calls={r["Calls"]:>5s} avg_us={float(r["AverageNs"])/1e3:9.1f}')
d = json.loads(open("$OUT/trace.json").read().strip().splitlines()[-1])
print("qps", round(d["value"]), "ms/step", round(d["ms_per_step"], 4))
PY
