#!/bin/bash
# Shader clock and matrix-pipe occupancy of the main pass and of its timing skeletons (scripts/r04/skeletons.sh): counters and the
# kernel's duration from ONE run each (--pmc with --kernel-trace only).   Usage: scripts/r04/skeleton_clocks.sh OUTDIR [modes...]
OUT=${1:-gpurun_out/r04skc}; shift
MODES=${@:-0 31 24 17 18 20 0}
mkdir -p $OUT; export TMPDIR=/tmp
cp frankensearch_amd/libfsgpu.so /tmp/libfsgpu_default.so
cp frankensearch_amd/libfsgpu_variant_exp.so frankensearch_amd/libfsgpu.so
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-two-tier"
i=0
for dbg in $MODES; do
  i=$((i+1))
  FSGPU_WIDE_DBG=$dbg timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES \
      --kernel-trace --output-format csv -d $OUT/run${i}_dbg$dbg -o bench -- $B > $OUT/run${i}_dbg$dbg.log 2>&1
  python - <<PY | tee -a $OUT/skeleton_clocks.txt
import csv, glob, collections
dur = collections.defaultdict(list); ctr = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/run${i}_dbg$dbg/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "scan_wide_kernel<384, 1, 4, 3, " in r["Kernel_Name"]:
            dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
for f in glob.glob("$OUT/run${i}_dbg$dbg/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "scan_wide_kernel<384, 1, 4, 3, " in r["Kernel_Name"]:
            ctr[r["Dispatch_Id"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
rows = []
for d, c in ctr.items():
    if d in dur:
        v = {k: sum(x) for k, x in c.items()}
        rows.append((dur[d], v))
rows = rows[len(rows) // 3:]   # past the warm-up
if rows:
    ms = sum(r[0] for r in rows) / len(rows)
    g = lambda k: sum(r[1].get(k, 0.0) for r in rows) / len(rows)
    gui = g("GRBM_GUI_ACTIVE") / 8.0   # summed over the 8 XCDs
    print(f"dbg=$dbg launches={len(rows)} avg_ms={ms:.4f} clock_GHz={gui / ms / 1e6:.3f} mfma_busy={g('SQ_VALU_MFMA_BUSY_CYCLES') / (gui * 1024):.3f} "
          f"wait_any/wave={g('SQ_WAIT_ANY') / max(g('SQ_WAVE_CYCLES'), 1):.3f} wait_inst/wave={g('SQ_WAIT_INST_ANY') / max(g('SQ_WAVE_CYCLES'), 1):.3f} "
          f"active_inst/wave={g('SQ_ACTIVE_INST_ANY') / max(g('SQ_WAVE_CYCLES'), 1):.3f} wave_cycles_per_wave={g('SQ_WAVE_CYCLES') / (256 * 8 * 2):.0f}")
PY
done
cp /tmp/libfsgpu_default.so frankensearch_amd/libfsgpu.so
