#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/enclong; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -o enc -- python scripts/ubench/enc_long.py > $O/log.txt 2>&1
grep "bert " $O/log.txt
python - <<PY
import csv
for r in csv.DictReader(open("$O/t/enc_kernel_stats.csv")):
    n = r["Name"]
    if "bert" in n and "pack" not in n and "to_half" not in n: print("   %-60s calls %5s avg %8.1f min %8.1f" % (n.replace("_ZN5fsgpu","")[:60], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3))
PY
