#!/usr/bin/env python3
"""MFMA utilisation of the encoder kernels from two rocprofv3 runs of scripts/bench_encoders.py:

    rocprofv3 --kernel-trace --output-format csv -d <trace_dir> -- python scripts/bench_encoders.py
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d <pmc_dir> -- python scripts/bench_encoders.py
    python scripts/encoder_pmc_summary.py <trace_dir> <pmc_dir> out.json

mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (kernel duration x 2.4 GHz x 1024 SIMDs): the fraction of SIMD-cycles the matrix
pipe was busy while the kernel ran (MfmaUtil as the gfx94x derived metric defines it; rocprofv3 ships no gfx950 derived
metrics, MI355X_MICROARCH.md).  The duration comes from the kernel trace of the same command (GRBM_GUI_ACTIVE is summed
over the eight XCDs on this part and is listed only for reference).  A v_mfma_f32_16x16x32_f16 keeps the pipe busy 16
cycles for 16,384 flops, so mfma_util x 2.5 PFLOP/s is the delivered f16 matrix rate."""
import csv
import glob
import json
import sys
from collections import defaultdict

trace_dir, pmc_dir, out_path = sys.argv[1:4]
dur = defaultdict(list)
for f in glob.glob(f"{trace_dir}/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[(r["Kernel_Name"], r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
ctr = defaultdict(lambda: defaultdict(list))
for f in glob.glob(f"{pmc_dir}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ctr[(r["Kernel_Name"], int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = []
for (name, gx, gy, gz), d in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    if "fsgpu" not in name or "bert" not in name:
        continue
    c = ctr.get((name, int(gx) * int(gy or 1) * int(gz or 1))) or {}
    busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", [])
    act = c.get("GRBM_GUI_ACTIVE", [])
    e = {"kernel": name, "grid": [int(gx), int(gy or 1), int(gz or 1)], "dispatches": len(d), "avg_us": sum(d) / len(d)}
    if busy and act:
        b, a = sum(busy) / len(busy), sum(act) / len(act)
        avg_s = e["avg_us"] * 1e-6
        e.update({"mfma_busy_cycles": b, "gui_active_cycles_all_xcds": a, "mfma_util": b / (avg_s * 2.4e9 * 1024),
                  "delivered_tflops_f16": b / 16.0 * 16384.0 / avg_s / 1e12})
    out.append(e)
json.dump(out, open(out_path, "w"), indent=1)
for e in out[:14]:
    print(f'{e["avg_us"]:9.1f} us x{e["dispatches"]:4d}  util={e.get("mfma_util", float("nan")):.3f}  grid={e["grid"]}  {e["kernel"][:70]}')
