#!/bin/bash
# Where the wide main pass's time goes, without a thread trace (this image ships no ATT decoder library): (1) SQ counters that
# split a wave's issue time by instruction type and name the FIFO stalls, (2) kernel-trace times of the timing skeletons
# (experiments build: FSGPU_WIDE_DBG 1 = no MFMAs, 2 = no DMA, 4 = no tile barrier / DMA wait, 5 = no fragment reads).
# Usage: scripts/r03/diag_wide.sh OUTDIR OPT
OUT=${1:-gpurun_out/diag}; OPT=${2:-7}
mkdir -p $OUT; export TMPDIR=/tmp; export FSGPU_WIDE_OPT=$OPT
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-two-tier"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_VMEM \
    --output-format csv -d $OUT/pmc_a -o bench -- $B > $OUT/pmc_a.log 2>&1
python scripts/pmc_summary.py $OUT/pmc_a $OUT/pmc_a.json | grep "scan_wide_kernel<384, 1, 4, 6, $OPT, 0>" | sed "s/^/A /"
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_IFETCH SQ_INSTS_BRANCH SQ_WAIT_ANY \
    --output-format csv -d $OUT/pmc_b -o bench -- $B > $OUT/pmc_b.log 2>&1
python scripts/pmc_summary.py $OUT/pmc_b $OUT/pmc_b.json | grep "scan_wide_kernel<384, 1, 4, 6, $OPT, 0>" | sed "s/^/B /"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM GRBM_GUI_ACTIVE \
    --output-format csv -d $OUT/pmc_c -o bench -- $B > $OUT/pmc_c.log 2>&1
python scripts/pmc_summary.py $OUT/pmc_c $OUT/pmc_c.json | grep "scan_wide_kernel<384, 1, 4, 6, $OPT, 0>" | sed "s/^/C /"
for dbg in 0 1 2 4 5; do
  FSGPU_WIDE_OPT=7 FSGPU_WIDE_DBG=$dbg rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_dbg$dbg -o bench -- $B > $OUT/trace_dbg$dbg.log 2>&1
  python - <<PY
import csv, glob
for f in glob.glob("$OUT/trace_dbg$dbg/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "scan_wide_kernel<384, 1, 4, 6, 7, " in r["Name"] and not r["Name"].split("(")[0].endswith(", 3>"):
            print(f'skeleton dbg=$dbg {r["Name"][:60]} calls={r["Calls"]} avg_ms={float(r["AverageNs"])/1e6:.4f}')
PY
done
