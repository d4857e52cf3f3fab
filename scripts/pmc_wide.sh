#!/bin/bash
# SQ counters of the wide main pass under env assignments.  Usage: scripts/pmc_wide.sh TAG [ENV=VAL...]
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE \
    --output-format csv -d $OUT/pmc_sq -o bench -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-two-tier > $OUT/pmc_sq.log 2>&1
python scripts/pmc_summary.py $OUT/pmc_sq $OUT/pmc_sq.json | grep -i "scan_wide" | sed "s/^/$TAG /"
