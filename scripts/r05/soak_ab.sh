#!/bin/bash
# Same-box soak of filtered batched searches (scripts/r04/bitmap_soak.py) on three libraries: the r04 failing configuration (one scalar
# read of a bitmap word, no per-wave s_dcache_inv), the double read without the invalidate, and the shipped build (double read + invalidate).
#   scripts/r05/soak_ab.sh OUTDIR SECONDS_PER_VARIANT
O=${1:-gpurun_out/r05soak}; T=${2:-200}; mkdir -p $O; export TMPDIR=/tmp
L=frankensearch_amd/libfsgpu.so
cp $L /tmp/libfsgpu_default.so
{
echo "box: $(hostname) $(cat /sys/class/drm/card*/device/unique_id 2>/dev/null | head -1)"
for v in noinv_single noinv_double default; do
  [ $v = default ] && cp /tmp/libfsgpu_default.so $L || cp frankensearch_amd/libfsgpu_variant_$v.so $L
  echo "== $v"; python scripts/r04/bitmap_soak.py $T 100 2>&1 | grep -v amdgpu.ids | tail -n 6
done
cp /tmp/libfsgpu_default.so $L
} 2>&1 | tee $O/soak_ab_$(date +%s).txt
