#!/bin/bash
# SQ counters + kernel trace of the batched main pass (run on the GPU box through gpurun).  Counter passes are separate
# --pmc-only runs.  Usage: scripts/prof_wide.sh <tag> [extra env assignments...]
TAG=${1:-wide}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS \
    --output-format csv -d $OUT/pmc_sq -o bench -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-two-tier > $OUT/pmc_sq.log 2>&1
python scripts/pmc_summary.py $OUT/pmc_sq $OUT/pmc_sq.json | grep -i "scan_wide\|scan_mfma_kernel<384, 8, 8, 2" 
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-two-tier > $OUT/trace.json 2> $OUT/trace.err
python - <<PY
import csv, glob
for f in glob.glob("$OUT/trace/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    for r in rows[:8]:
        print(f'{r["Name"][:70]:70s} calls={r["Calls"]:>6s} avg_us={float(r["AverageNs"])/1e3:9.1f} total_ms={float(r["TotalDurationNs"])/1e6:9.2f} pct={r["Percentage"]}')
PY
