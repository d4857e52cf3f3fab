#!/usr/bin/env python3
"""GPU busy fraction and per-kernel time shares of the LAST `window` fraction of a rocprofv3 kernel trace (union of kernel intervals:
kernels of different streams overlap).  usage: gpu_busy.py <kernel_trace.csv> [window=0.4]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
win = float(sys.argv[2]) if len(sys.argv) > 2 else 0.4
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
t_end = max(e for _, e, _ in iv)
t_begin = min(s for s, _, _ in iv)
t0 = t_end - int((t_end - t_begin) * win)
iv = [x for x in iv if x[0] >= t0]
busy, cur_s, cur_e = 0, None, None
for s, e, _ in iv:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
span = iv[-1][1] - iv[0][0]
print(f"window {span / 1e6:.2f} ms, GPU busy (union) {busy / 1e6:.2f} ms = {busy / span:.3f}; sum of kernel durations {sum(e - s for s, e, _ in iv) / 1e6:.2f} ms")
by = defaultdict(lambda: [0, 0])
for s, e, n in iv:
    key = n.split("(")[0][:90]
    by[key][0] += e - s
    by[key][1] += 1
for k, (t, c) in sorted(by.items(), key=lambda x: -x[1][0])[:28]:
    print(f"{t / 1e6:9.2f} ms {100 * t / span:5.1f}%  x{c:6d}  avg {t / c / 1e3:8.1f} us  {k}")
