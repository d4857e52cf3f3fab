#!/bin/bash
# where the 64-query LDS-query shape stops paying: per-tier stage times at 2..96 queries, default build
for b in 2 8 16 32 48 64 65 96; do
  echo "B=$b"; B=$b REPS=20 python scripts/r06/prof_two_tier_stages.py 2>/dev/null | grep -E "two-pass|quality tier batched exact|fast tier batched exact"
done
