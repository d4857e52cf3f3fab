#!/bin/bash
# GPU session 5 of round 4: suite on the current build, the default line (two-tier p50s with the certified lone-query path), config 5 at
# 50M with the device-resident hand-off, shard step trace, the gloo rehearsal of --gpus 2
O=gpurun_out/r04s5; mkdir -p $O; export TMPDIR=/tmp
( time python -m pytest tests -m gpu -q -x ) > $O/pytest_all.txt 2>&1; tail -6 $O/pytest_all.txt
python bench.py --steps 20 --warmup 5 --no-adversarial > $O/bench_q.json 2> $O/bench_q.err
python - <<PY
import json
d = json.loads(open("$O/bench_q.json").read().strip().splitlines()[-1])
print("qps", round(d["value"]), "ms/step", round(d["ms_per_step"], 4), "main", round(d["roofline"]["avg_launch_ms"], 4), "frac", round(d["roofline"]["frac"], 4))
print("p50 single", d.get("p50_latency_ms_single_query"), "p50 phase0", d.get("p50_phase0_latency_ms"), "phase1", d.get("p50_phase1_latency_ms"), "spec", d.get("p50_phase1_latency_speculative_ms"), "rescored", d.get("p50_phase1_latency_rescored_fast_pool_ms"), "e2e", d.get("end_to_end_queries_per_sec"))
print("breakdown", d["two_tier"]["sequential_breakdown_ms"])
print("cpu", {k: d["cpu_baseline"][k] for k in ("value", "cores", "GBps", "parity_bit_exact", "batched_path_equals_oracle_8_queries")})
print("exact roofline", d["roofline"]["exact_f16_scan"]["frac"])
print("int8 two pass p50", d["int8_two_pass"]["p50_latency_ms"], "enc", d["encoders"]["minilm_l6"]["gpu_ms_per_batch"], d["encoders"]["minilm_l6"]["gpu_single_text_p50_ms"])
PY
python bench.py --rows 50000000 --config5 --no-cpu-baseline --no-two-tier --no-adversarial --no-encoders 2>/dev/null | tail -1 > $O/bench_config5.json
python - <<PY
import json
c = json.loads(open("$O/bench_config5.json").read().strip().splitlines()[-1])
print("config5 50M scan qps", round(c["value"]))
for k, v in c["config5"].items():
    print("  ", k, v if not isinstance(v, dict) else {x: (round(y, 3) if isinstance(y, float) else y) for x, y in v.items()})
PY
python bench.py --rows 1000000 --steps 30 --warmup 5 --no-cpu-baseline --no-two-tier --no-adversarial --no-encoders 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('1M: qps=%.0f step=%.3fms p50 single=%s' % (d['value'], d['ms_per_step'], d.get('p50_latency_ms_single_query')))"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_shard -o bench -- python bench.py --rows 1250000 --steps 30 --warmup 5 --no-cpu-baseline --no-two-tier --no-adversarial --no-encoders > $O/bench_shard_traced.json 2> $O/trace_shard.err
head -14 $O/trace_shard/*kernel_stats.csv | cut -c1-200
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_10m -o bench -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-two-tier --no-adversarial --no-encoders > $O/bench_10m_traced.json 2> $O/trace_10m.err
head -14 $O/trace_10m/*kernel_stats.csv | cut -c1-200
( time FSGPU_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 10 --warmup 3 --config5-rows 6000000 ) > $O/rehearsal_gloo2.txt 2>&1; tail -c 1500 $O/rehearsal_gloo2.txt
