#!/bin/bash
# Timing skeletons of the shipped main pass (scan_wide_kernel<384, 1, 4, 3, 30, D>, experiments build of mfma_wide.hip: D = 16 + bits,
# 1 no DMA, 2 no per-tile wait + barrier, 4 no fragment reads in the loop, 8 no threshold tests): kernel-trace averages on ONE box.
# The skeletons' answers are not valid — only their durations are read.   Usage: scripts/r04/skeletons.sh OUTDIR [modes...]
OUT=${1:-gpurun_out/r04sk}; shift
MODES=${@:-0 17 18 20 24 19 22 23 26 31 0}
mkdir -p $OUT; export TMPDIR=/tmp
cp frankensearch_amd/libfsgpu.so /tmp/libfsgpu_default.so
cp frankensearch_amd/libfsgpu_variant_exp.so frankensearch_amd/libfsgpu.so
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-two-tier"
for dbg in $MODES; do
  FSGPU_WIDE_DBG=$dbg timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_dbg$dbg -o bench -- $B > $OUT/trace_dbg$dbg.log 2>&1
  python - <<PY | tee -a $OUT/skeletons.txt
import csv, glob
for f in glob.glob("$OUT/trace_dbg$dbg/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Name"].split("(")[0]
        if "scan_wide_kernel<384, 1, 4, 3, 30, " in n:
            print(f'dbg=$dbg {n[-40:]} calls={r["Calls"]} avg_ms={float(r["AverageNs"])/1e6:.4f} min_ms={float(r["MinNs"])/1e6:.4f}')
PY
done
cp /tmp/libfsgpu_default.so frankensearch_amd/libfsgpu.so
