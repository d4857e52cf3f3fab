#!/bin/bash
# LDS-query main-pass shapes for int8 rows on batches of 65..255 queries: shape 4 (64-row tiles, 128 accumulator registers: spills inside the
# tile loop) against shapes 2 and 1 (32- / 16-row tiles, no spills) — experiments build of the planner (FSGPU_MFMA_SHAPE_I8)
L=frankensearch_amd/libfsgpu.so
cp $L /tmp/libfsgpu_default.so
cp frankensearch_amd/libfsgpu_variant_exp.so $L
for b in 100 128 200 255; do
  for sh in 4 2 1; do
    echo "B=$b shape=$sh"; B=$b REPS=20 FSGPU_MFMA_SHAPE_I8=$sh python scripts/r06/prof_two_tier_stages.py 2>/dev/null | grep -E "two-pass|quality tier batched exact \(fetch 30\)"
  done
done
cp /tmp/libfsgpu_default.so $L
