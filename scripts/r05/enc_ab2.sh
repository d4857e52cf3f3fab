#!/bin/bash
# Same-box A/B of the 32-row post-attention block's loop forms (variants built by scripts/r04/build_variant.sh, all with the 64-row form off).
O=${1:-gpurun_out/r05encab2}; mkdir -p $O; export TMPDIR=/tmp
L=frankensearch_amd/libfsgpu.so
cp $L /tmp/libfsgpu_default.so
{
for v in rows32 a_fixed_nosched b_rolled_sched c_rolled_nosched; do
  cp frankensearch_amd/libfsgpu_variant_$v.so $L
  for rep in 1 2 3; do printf "%-18s " $v; WARM=100 N=400 python scripts/r05/enc_docs_only.py 2>&1 | tail -n 1; done
done
cp /tmp/libfsgpu_default.so $L
for rep in 1 2 3; do printf "%-18s " default; WARM=100 N=400 python scripts/r05/enc_docs_only.py 2>&1 | tail -n 1; done
} 2>&1 | tee $O/enc_ab2.txt
