mkdir -p gpurun_out/s3
python scripts/fuzz_batched.py 21 120 > gpurun_out/s3/fuzz_b.log 2>&1; tail -3 gpurun_out/s3/fuzz_b.log
B="python bench.py --no-cpu-baseline --no-two-tier --steps 20 --warmup 3"
for wm in 3 4; do for gr in 2 4 8; do
  FSGPU_WIDE_MAX=$wm FSGPU_I8F_GROWTH=$gr $B > gpurun_out/s3/b_wm${wm}_g${gr}.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("gpurun_out/s3/b_wm${wm}_g${gr}.json")); r=d["roofline"]
print("wm=$wm growth=$gr", round(d["value"]), round(d["ms_per_step"],3), "launch_ms", round(r["avg_launch_ms"],4), r["launches"], d["config"]["exact_fallback_queries"])
PY
done; done
FSGPU_ROUND=1280 $B --batch 1280 > gpurun_out/s3/b_1280.json 2>/dev/null
python - <<PY
import json
d=json.load(open("gpurun_out/s3/b_1280.json")); r=d["roofline"]
print("batch 1280 (5+5)", round(d["value"]), round(d["ms_per_step"],3), "launch_ms", round(r["avg_launch_ms"],4), r["launches"], d["config"]["exact_fallback_queries"])
PY
