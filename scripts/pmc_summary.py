#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc run (counter_collection.csv) per kernel: mean counter value per dispatch.

    python scripts/pmc_summary.py <dir with *_counter_collection.csv> [out.json]

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB; on gfx950 FETCH_SIZE counts wide coalesced reads at half
their size (MI355X_MICROARCH.md, HBM section), so `hbm_read_bytes_corrected` = FETCH_SIZE * 1024 * 2.
"""
import csv
import glob
import json
import sys
from collections import defaultdict

d = sys.argv[1]
files = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)
acc = defaultdict(list)
for f in files:
    for r in csv.DictReader(open(f)):
        acc[(r["Kernel_Name"], r["Counter_Name"], r.get("Grid_Size", ""))].append(float(r["Counter_Value"]))
out = []
for (kern, ctr, grid), vals in sorted(acc.items()):
    e = {"kernel": kern, "counter": ctr, "grid": grid, "dispatches": len(vals), "avg": sum(vals) / len(vals),
         "min": min(vals), "max": max(vals)}
    if ctr == "FETCH_SIZE":
        e["hbm_read_bytes_corrected"] = e["avg"] * 1024 * 2
    if ctr == "WRITE_SIZE":
        e["hbm_write_bytes_uncalibrated"] = e["avg"] * 1024
    out.append(e)
txt = json.dumps(out, indent=1)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(txt)
for e in out:
    if "fsgpu" in e["kernel"]:
        print(f'{e["kernel"][:80]:80s} {e["counter"]:12s} n={e["dispatches"]:4d} avg={e["avg"]:.1f}')
