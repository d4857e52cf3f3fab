#!/bin/bash
# s_memtime stamps of the shipped main pass (experiments build, FSGPU_WIDE_DBG=8): per launch, the share of a wave's tile-loop cycles
# spent between arriving at a tile's DMA wait + barrier and leaving it, and inside the append path — the 10M bench shape and a 1.25M-row shard.
O=${1:-gpurun_out/r04stamps}; mkdir -p $O
cp frankensearch_amd/libfsgpu.so /tmp/base.so
cp frankensearch_amd/libfsgpu_variant_exp.so frankensearch_amd/libfsgpu.so
for rows in 10000000 1250000; do
  echo "== $rows rows x 384, 1,024 queries per launch (two 512-query groups)"
  FSGPU_WIDE_DBG=8 python bench.py --rows $rows --steps 8 --warmup 3 --blocking-steps --no-cpu-baseline --no-two-tier --no-adversarial --no-encoders 2>&1 >/dev/null | grep "wide stamps" | tail -6
done | tee $O/wide_stamps.txt
cp /tmp/base.so frankensearch_amd/libfsgpu.so
