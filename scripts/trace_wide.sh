#!/bin/bash
# kernel-trace average of the wide main pass under the given env assignments.  Usage: scripts/trace_wide.sh TAG [ENV=VAL...] [-- bench args]
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1"
while [ $# -gt 0 ]; do if [ "$1" == "--" ]; then shift; ARGS="$@"; break; fi; export "$1"; shift; done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python bench.py --no-cpu-baseline --no-two-tier $ARGS > $OUT/trace.json 2> $OUT/trace.err
python - <<PY
import csv, glob
for f in glob.glob("$OUT/trace/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    for r in rows[:12]:
        if "scan_wide" in r["Name"] or "scan_mfma" in r["Name"] or "select" in r["Name"]:
            print(f'$TAG {r["Name"][:64]:64s} calls={r["Calls"]:>5s} avg_us={float(r["AverageNs"])/1e3:9.1f}')
PY
