#!/bin/bash
# Same-box A/B of the attention kernel's round-5 changes on 32 documents x 512 tokens (variants: scripts/r04/build_variant.sh NAME bert_kernels.hip "-D...").
O=${1:-gpurun_out/r05attn}; mkdir -p $O; export TMPDIR=/tmp
L=frankensearch_amd/libfsgpu.so
cp $L /tmp/libfsgpu_default.so
{
for v in default z4 noperm; do
  [ $v = default ] && cp /tmp/libfsgpu_default.so $L || cp frankensearch_amd/libfsgpu_variant_$v.so $L
  for rep in 1 2; do printf "%-10s " $v; WARM=100 N=400 python scripts/r05/enc_docs_only.py 2>&1 | tail -n 1; done
done
cp /tmp/libfsgpu_default.so $L
} 2>&1 | tee $O/attn_ab.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/enc_trace -o enc -- python scripts/r05/enc_docs_only.py > $O/enc_traced.log 2>&1
head -5 $O/enc_trace/*kernel_stats.csv | cut -c1-150 | tee $O/enc_kernels.txt
