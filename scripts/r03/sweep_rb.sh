#!/bin/bash
# lab build (FSGPU_BUILD_DEFS=-DFSGPU_EXPERIMENTS): size of the second sample (FSGPU_RB) and rank of the heuristic gate (FSGPU_HEUR_RANK)
# against queries/s, main-pass time and fallbacks at the default bench shape
export FSGPU_BUILD_DEFS="-DFSGPU_EXPERIMENTS"
run() { python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-two-tier --no-adversarial --no-encoders 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('qps=%.0f step=%.3fms main=%.4fms fb=%s refiltered=%s' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['config'].get('exact_fallback_queries'), d['config'].get('refiltered_f16_queries')))"; }
echo -n "default: "; run
for rb in 131072 196608 262144 524288 786432; do echo -n "RB=$rb: "; FSGPU_RB=$rb run; done
for hr in 4 6 12 16; do echo -n "HEUR_RANK=$hr: "; FSGPU_HEUR_RANK=$hr run; done
for sm in 8 12; do echo -n "SLOTS_MAIN=$sm: "; FSGPU_SLOTS_MAIN=$sm run; done
echo -n "default again: "; run
