export TMPDIR=/tmp
O=gpurun_out/shard; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -o b -- python bench.py --rows 1250000 --no-cpu-baseline --no-two-tier > $O/bench.json 2> $O/err.txt
python - <<PY
import csv
rows = list(csv.DictReader(open("$O/t/b_kernel_stats.csv")))
for r in rows[:14]:
    print("%-70s calls %5s avg %9.1f us  total %8.2f ms" % (r["Name"].replace("fsgpu::","")[:70], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
tail -c 300 $O/bench.json
