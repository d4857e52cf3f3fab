#!/bin/bash
# timeline of single-query MiniLM forwards: kernel time and the gaps between the kernels of one replayed graph
export TMPDIR=/tmp
O=gpurun_out/encsingle; mkdir -p $O
cat > /tmp/enc_single.py <<PY
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import frankensearch_amd as fa
from frankensearch_amd.synthetic import random_bert_weights
bert = fa.NativeEmbedder(random_bert_weights(1, 30522, 384, 6, 1536))
ids = np.asarray([101] + list(range(2000, 2014)) + [102], dtype=np.int32); offs = np.asarray([0, ids.size], dtype=np.uint32)
out = np.empty((1, 384), dtype=np.float32)
for _ in range(5): bert.embed_flat(ids, offs, out)
t0 = time.perf_counter()
for _ in range(100): bert.embed_flat(ids, offs, out)
print("single %.3f ms" % ((time.perf_counter() - t0) / 100 * 1e3))
PY
rocprofv3 --kernel-trace --output-format csv -d $O/t -o enc -- python /tmp/enc_single.py > $O/log.txt 2>&1
grep single $O/log.txt
python - <<PY
import csv
rows = sorted(csv.DictReader(open("$O/t/enc_kernel_trace.csv")), key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if "bert_q" in r["Kernel_Name"]]
# group into forwards of 25 kernels; take the last 20 forwards
n = 25
fw = [rows[i:i + n] for i in range(0, len(rows) - n + 1, n)][-20:]
import statistics
tot = [(int(f[-1]["End_Timestamp"]) - int(f[0]["Start_Timestamp"])) / 1e3 for f in fw]
ker = [sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in f) / 1e3 for f in fw]
print("first kernel start -> last kernel end: median %.1f us; kernel time %.1f us; gaps %.1f us" % (statistics.median(tot), statistics.median(ker), statistics.median(tot) - statistics.median(ker)))
f = fw[-1]
for a, b in zip(f[:9], f[1:10]):
    print("  %-40s %6.1f us   gap to next %5.1f us" % (a["Kernel_Name"][:40], (int(a["End_Timestamp"]) - int(a["Start_Timestamp"])) / 1e3, (int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3))
PY
