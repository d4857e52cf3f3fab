run() { python bench.py --rows $ROWS --steps 40 --warmup 5 --no-cpu-baseline --no-two-tier 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('qps=%.0f step=%.3fms main=%.4fms fb=%s' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['config']['exact_fallback_queries']))"; }
for ROWS in 1250000 2500000; do
  echo "rows=$ROWS default: $(run)"
  for ra in 4096 8192; do for rb in 16384 32768 65536 131072; do echo "rows=$ROWS RA=$ra RB=$rb: $(FSGPU_RA=$ra FSGPU_RB=$rb run)"; done; done
done
