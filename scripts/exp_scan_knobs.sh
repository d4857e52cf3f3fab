run() { python bench.py --exact --steps 100 --warmup 5 --no-cpu-baseline --no-two-tier "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('qps=%.0f step=%.3fms scan=%.4fms frac=%.3f' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac']))"; }
echo "B1 nt"; run --batch 1
echo "B1 plain"; run --batch 1 --variant 2
echo "B2 nt"; run --batch 2
echo "B2 plain"; run --batch 2 --variant 2
for g in 256 512 768 1024 2048; do echo "B1 grid $g"; FSGPU_GRID_BLOCKS=$g run --batch 1; done
for g in 256 768 1024; do echo "B2 grid $g"; FSGPU_GRID_BLOCKS=$g run --batch 2; done
echo "B4"; run --batch 4
