#!/bin/bash
# A/B of the wide main pass's options (mfma_wide.hip OPT bits: 1 = lag, 2 = neg-tau, 4 = saddr) on the bench shape, same box, same
# process sequence, interleaved twice.  Needs a build with FSGPU_BUILD_DEFS="-DFSGPU_EXPERIMENTS ..." (scripts/r03/README).
# Usage: scripts/r03/ab_wide_opt.sh OUTDIR "opt list" [rounds]
OUT=${1:-gpurun_out/ab}; OPTS=${2:-"0 1 2 3 4 5 6 7"}; ROUNDS=${3:-2}
mkdir -p $OUT
for r in $(seq 1 $ROUNDS); do
  for o in $OPTS; do
    FSGPU_WIDE_OPT=$o python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-two-tier --no-adversarial --no-encoders > $OUT/opt${o}_r$r.json 2> $OUT/opt${o}_r$r.err
    python - <<PY
import json
try:
    d = json.loads(open("$OUT/opt${o}_r$r.json").read().strip().splitlines()[-1])
    rf = d["roofline"]
    print(f"opt=$o round=$r qps={d['value']:.0f} ms_per_step={d['ms_per_step']:.3f} main_pass_ms={rf['avg_launch_ms']:.4f} frac={rf['frac']:.4f} fallbacks={d['config'].get('exact_fallback_queries')}")
except Exception as e:
    print("opt=$o round=$r FAILED", e)
PY
  done
done | tee $OUT/summary.txt
for dbg in 1 2; do
  FSGPU_WIDE_OPT=7 FSGPU_WIDE_DBG=$dbg python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-two-tier --no-adversarial --no-encoders > $OUT/dbg$dbg.json 2> $OUT/dbg$dbg.err
  python - <<PY | tee -a $OUT/summary.txt
import json
try:
    d = json.loads(open("$OUT/dbg$dbg.json").read().strip().splitlines()[-1])
    print(f"skeleton dbg=$dbg (1 = no MFMAs, 2 = no DMA; answers invalid) main_pass_ms={d['roofline']['avg_launch_ms']:.4f}")
except Exception as e:
    print("dbg=$dbg FAILED", e, open("$OUT/dbg$dbg.err").read()[-400:])
PY
done
