#!/bin/bash
# per-tier stage times of small batches (default build): where the 64-query shape, the 128-slot shape and the padded 256-slot pass cross
for b in ${BS:-16 32 48 64 65 80 96 128 129}; do
  echo "B=$b"; B=$b REPS=20 python scripts/r06/prof_two_tier_stages.py 2>/dev/null | grep -E "two-pass|quality tier batched exact|fast tier batched exact"
done
