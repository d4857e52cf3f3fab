#!/bin/bash
# 129..255 queries: one padded 256-slot pass of the register-resident-query kernel (default since round 6) against what rounds 3-5 did
# (128 + the rest on the LDS-query kernel); and the per-tier stage times around the boundary, default build.
mkdir -p gpurun_out/r06
for b in 64 100 128 129 160 200 255 256 300 384; do
  echo "B=$b"; B=$b REPS=20 python scripts/r06/prof_two_tier_stages.py 2>/dev/null | grep -E "two-pass|quality tier batched exact \(fetch 30\)"
done
