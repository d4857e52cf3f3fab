#!/bin/bash
# Same-box A/B of the encoder's large-M post-attention block: the shipped library (64-row blocks above 8,192 tokens) against the
# variant that keeps 32-row blocks (scripts/r04/build_variant.sh rows32 bert_gemm_w.hip "-DFSGPU_FFN_W64_MIN_ROWS=1073741824"),
# and against the round-4 build when its library is present.   scripts/r05/enc_ab.sh OUTDIR
O=${1:-gpurun_out/r05encab}; mkdir -p $O; export TMPDIR=/tmp
L=frankensearch_amd/libfsgpu.so
cp $L /tmp/libfsgpu_default.so
run() {   # name library
  cp $2 $L
  for rep in 1 2; do printf "%-8s " $1; N=30 python scripts/r05/enc_docs_only.py 2>&1 | tail -n 1; done
  printf "%-8s " $1; python scripts/bench_encoders.py 2>&1 | grep "batch 256 (5179"
}
{
run default /tmp/libfsgpu_default.so
run rows32 frankensearch_amd/libfsgpu_variant_rows32.so
[ -f frankensearch_amd/libfsgpu_variant_r04.so ] && run r04 frankensearch_amd/libfsgpu_variant_r04.so
cp /tmp/libfsgpu_default.so $L
} 2>&1 | tee $O/enc_ab.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/enc_trace -o enc -- python scripts/r05/enc_docs_only.py > $O/enc_traced.log 2>&1
head -6 $O/enc_trace/*kernel_stats.csv | cut -c1-150 | tee $O/enc_kernels.txt
