#!/bin/bash
# how many appends does the main pass make, and how does that move with the second sample's size?  WRITE_SIZE (64-byte transactions, one
# per 8-byte append) + the kernel's duration per FSGPU_RB_PCT (variant "expvi").   scripts/r04/rb_writes.sh OUTDIR
O=${1:-gpurun_out/r04rbw}; mkdir -p $O; export TMPDIR=/tmp
cp frankensearch_amd/libfsgpu.so /tmp/base.so
cp frankensearch_amd/libfsgpu_variant_expvi.so frankensearch_amd/libfsgpu.so
for pct in 100 200 400; do
  FSGPU_RB_PCT=$pct rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pct$pct -o bench -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-two-tier --no-adversarial --no-encoders > $O/pct$pct.log 2>&1
  python - <<PY | tee -a $O/rb_writes.txt
import csv, glob, collections
dur = {}; w = collections.defaultdict(float); names = {}
for f in glob.glob("$O/pct$pct/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "scan_wide_kernel<384, 1, 4," in r["Kernel_Name"]:
            dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6; names[r["Dispatch_Id"]] = r["Kernel_Name"].split("(")[0][-30:]
for f in glob.glob("$O/pct$pct/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Dispatch_Id"] in dur and r["Counter_Name"] == "WRITE_SIZE": w[r["Dispatch_Id"]] += float(r["Counter_Value"])
by = collections.defaultdict(list)
for d in dur: by[names[d]].append((dur[d], w[d]))
for n, v in by.items():
    v = v[len(v) // 3:]
    print(f"pct=$pct {n}: launches={len(v)} avg_ms={sum(x[0] for x in v)/len(v):.4f} WRITE_SIZE_KiB={sum(x[1] for x in v)/len(v):.0f} -> appends ~{sum(x[1] for x in v)/len(v)*1024/64/1e6:.2f} M per launch")
PY
done
cp /tmp/base.so frankensearch_amd/libfsgpu.so
