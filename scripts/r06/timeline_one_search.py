"""Prints the kernel timeline of the LAST quality-tier search (int8-filtered exact, 384 dimensions) found in a rocprofv3 kernel trace of
scripts/r06/prof_two_tier_stages.py: what a small batch spends outside its main pass."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "scan_mfma_kernel<384, 8, 8, 2, 2" in r["Kernel_Name"] or "scan_wide_kernel<384, 1, 4, 3, 30, 0>" in r["Kernel_Name"]]
i = idx[-1]
lo, hi = max(0, i - 6), min(len(rows), i + 6)
t0 = int(rows[lo]["Start_Timestamp"])
for r in rows[lo:hi]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{s / 1e3:9.1f} -> {e / 1e3:9.1f} us ({(e - s) / 1e3:7.1f})  {r['Kernel_Name'].replace('fsgpu::', '')[:100]}")
