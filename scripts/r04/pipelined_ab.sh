#!/bin/bash
# begin / end with one step enqueued ahead against one blocking call per step (bench.py --blocking-steps): the bench shape and a
# 1.25M-row shard at N = 1, the gloo rehearsal of --gpus 2 (the per-rank loop of frankensearch_amd/sharded.py), the tests of both halves.
O=${1:-gpurun_out/r04pipe}; mkdir -p $O; export TMPDIR=/tmp
( time python -m pytest tests/test_gpu_parity.py tests/test_gpu_sharded.py tests/test_gpu_int8_filter.py -m gpu -q -x ) > $O/pytest.txt 2>&1; grep -n "passed\|failed" $O/pytest.txt | tail -2
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  qps=%.0f step=%.4fms main=%.4fms frac=%.3f fb=%s  %s' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['config'].get('exact_fallback_queries'), d['config'].get('host_loop')))"; }
{
for mode in "" "--blocking-steps" "" "--blocking-steps"; do
  python bench.py $mode --steps 60 --warmup 10 --no-cpu-baseline --no-two-tier --no-adversarial --no-encoders 2>/dev/null | tail -1 | line
  echo "  shard 1.25M:"; python bench.py $mode --rows 1250000 --steps 100 --warmup 10 --no-cpu-baseline --no-two-tier --no-adversarial --no-encoders 2>/dev/null | tail -1 | line
done
} 2>&1 | tee $O/ab.txt
rocprofv3 --kernel-trace --output-format csv -d $O/trace -o bench -- python bench.py --rows 1250000 --steps 20 --warmup 5 --no-cpu-baseline --no-two-tier --no-adversarial --no-encoders > $O/bench_traced.json 2> $O/trace.err
python - <<PY
import csv, glob
f = glob.glob("$O/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "scan_wide_kernel<384, 1, 4, 3, 30, 0>" in r["Kernel_Name"]]
i = idx[len(idx) // 2]; prev = None
for r in rows[i - 4:i + 8]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f'{(s - prev) / 1e3 if prev else 0:7.1f} gap | {(e - s) / 1e3:8.1f} us | {r["Kernel_Name"][:70]}'); prev = e
PY
( time FSGPU_BENCH_BACKEND=gloo python bench.py --gpus 2 --rows 2500000 --steps 40 --warmup 5 --no-sharded-handle ) > $O/rehearsal_shard.txt 2>&1; grep "^{" $O/rehearsal_shard.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('gloo 2 ranks, 2.5M rows:', round(d['value']), 'q/s', round(d['ms_per_step'],4), 'ms/step', d['config'].get('host_loop'))"
