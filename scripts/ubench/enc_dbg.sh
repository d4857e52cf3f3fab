#!/bin/bash
# per-kernel durations of the batch-256 forward (argument: a label, e.g. 0).  The phase-ablation switch this script drove
# while the kernels were designed (FSGPU_GW_DBG: no stores / no MFMA loop / no GELU / no A loads / no W loads) was removed from
# the kernels afterwards; what it measured is recorded in DESIGN 3.5.
# the LN GEMM is reported separately for its two uses (attention output K = hidden, FFN down K = inter)
export TMPDIR=/tmp
O=gpurun_out/encdbg; mkdir -p $O
for d in "$@"; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/t$d -o enc -- python scripts/ubench/enc_batch.py > $O/log$d.txt 2>&1
  echo "== dbg $d: $(grep 'bert batch' $O/log$d.txt)"
  python - <<PY
import csv, collections
rows = sorted(csv.DictReader(open("$O/t$d/enc_kernel_trace.csv")), key=lambda r: int(r["Start_Timestamp"]))
st = collections.defaultdict(list); n_ln = 0
for r in rows:
    n = r["Kernel_Name"].replace("_ZN5fsgpu", "")
    if "bert" not in n or "pack" in n or "to_half" in n: continue
    key = n[:34]
    if "gemm_ln_w" in n or "ffn" in n:
        key += " #%d" % (n_ln % 2); n_ln += 1
    st[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in st.items():
    v.sort(); print("   %-40s calls %4d  median %6.1f  min %6.1f" % (k, len(v), v[len(v)//2], v[0]))
PY
done
