#!/bin/bash
# the finish's selection at fetch 30 (the two-tier flow): extraction scheme (ranks <= 32, shipped) against compaction + bitonic sort,
# variant library = mfma_scan.hip compiled with -DFSGPU_EXPERIMENTS (FSGPU_SELECT_SORT_ABOVE read from the environment)
L=frankensearch_amd/libfsgpu.so
V=frankensearch_amd/libfsgpu_variant_selexp.so
[ -f $V ] || { echo "(no variant library)"; exit 0; }
cp $L /tmp/libfsgpu_default.so
cp $V $L
for sa in 32 24 16 8; do
  for b in ${BS:-1024 64}; do
    echo "FSGPU_SELECT_SORT_ABOVE=$sa B=$b"; FSGPU_SELECT_SORT_ABOVE=$sa B=$b REPS=20 python scripts/r06/prof_two_tier_stages.py 2>/dev/null | grep -E "quality tier batched exact|fast tier batched exact"
  done
done
cp /tmp/libfsgpu_default.so $L
