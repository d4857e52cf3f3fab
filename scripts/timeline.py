"""Prints the GPU timeline of the last batched step from a rocprofv3 kernel trace csv (tuning aid)."""
import csv, sys
tr = list(csv.DictReader(open(sys.argv[1])))
tr.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(tr) if "prepare_queries" in r["Kernel_Name"]]
i0, i1 = idx[-2], idx[-1]
t0 = int(tr[i0]["Start_Timestamp"])
for r in tr[i0:i1]:
    print("%8.1f +%7.1fus  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Kernel_Name"][:70]))
print("step total: %.1f us" % ((int(tr[i1]["Start_Timestamp"]) - t0) / 1e3))
