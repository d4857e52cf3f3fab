#!/bin/bash
# rounds of 1,280 queries (two 640-query passes: 5 query tiles per wave) against the shipped 1,024 (two 512-query passes) for the
# many-queries engine's chunks — experiments build (FSGPU_ROUND)
L=frankensearch_amd/libfsgpu.so
V=frankensearch_amd/libfsgpu_variant_exp.so
[ -f $V ] || { echo "(no experiments library)"; exit 0; }
cp $L /tmp/libfsgpu_default.so
cp $V $L
echo "rounds of 1,024, chunks of 1,024"; CASES=0:1024:0 NQ=40960 python scripts/r06/exp_two_tier_many.py 2>&1 | grep "qps="
echo "rounds of 1,280, chunks of 1,280"; FSGPU_ROUND=1280 CASES=0:1280:0 NQ=40960 python scripts/r06/exp_two_tier_many.py 2>&1 | grep "qps="
echo "rounds of 1,280, chunks of 2,560"; FSGPU_ROUND=1280 CASES=0:2560:0 NQ=40960 python scripts/r06/exp_two_tier_many.py 2>&1 | grep "qps="
for b in 1024 1280; do echo "stages B=$b (FSGPU_ROUND=1280)"; FSGPU_ROUND=1280 B=$b REPS=10 python scripts/r06/prof_two_tier_stages.py 2>/dev/null | grep -E "two-pass|quality tier batched exact \(fetch 30\)|MiniLM"; done
cp /tmp/libfsgpu_default.so $L
