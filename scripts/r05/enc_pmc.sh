#!/bin/bash
# counters of the large-M encoder kernels (separate --pmc passes, no trace domains).  scripts/r05/enc_pmc.sh OUTDIR
O=${1:-gpurun_out/r05encpmc}; mkdir -p $O; export TMPDIR=/tmp
python scripts/r05/enc_docs_only.py | tee $O/untraced.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o enc -- python scripts/r05/enc_docs_only.py > $O/trace.log 2>&1
head -8 $O/trace/*kernel_stats.csv | cut -c1-150
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  N=3 rocprofv3 --pmc $set --output-format csv -d $O/pmc$i -o enc -- python scripts/r05/enc_docs_only.py > $O/pmc$i.log 2>&1
  python scripts/pmc_summary.py $O/pmc$i $O/pmc$i.json 2>/dev/null | grep -i "gemm_w\|ffn_w\|attention_lds" | cut -c1-50,80-140
done | tee $O/pmc_summary.txt
