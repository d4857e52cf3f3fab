#!/bin/bash
# production build on the GPU box: the whole GPU suite, the fuzzers, encoder micro-benchmarks, a quick default-shape bench line
O=gpurun_out/${1:-r03d}
mkdir -p $O
( time python -m pytest tests -m gpu -q ) > $O/pytest.txt 2>&1; grep -E "passed|failed|real" $O/pytest.txt; grep -E "^E  |^FAILED" $O/pytest.txt | head -20
python tests/fuzz_encoders.py 7 > $O/fuzz_enc.txt 2>&1; tail -1 $O/fuzz_enc.txt
python scripts/fuzz_batched.py 43 100 > $O/fuzz_batched.txt 2>&1; tail -1 $O/fuzz_batched.txt
python tests/fuzz_exact.py 11 > $O/fuzz_exact.txt 2>&1; tail -1 $O/fuzz_exact.txt
python scripts/bench_encoders.py > $O/enc.txt 2>&1; grep -E "^(m2v|bert)" $O/enc.txt
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-adversarial > $O/bench_q.json 2>/dev/null
python bench.py --rows 50000000 --config5 --no-cpu-baseline --no-two-tier --no-adversarial --no-encoders 2>/dev/null | tail -1 > $O/bench_config5.json
python - <<PY
import json
d = json.loads(open("$O/bench_q.json").read().strip().splitlines()[-1])
print("qps", round(d["value"]), "main", round(d["roofline"]["avg_launch_ms"], 4), "e2e", d.get("end_to_end_queries_per_sec"), "p50 phase1", d.get("p50_phase1_latency_ms"))
c = json.loads(open("$O/bench_config5.json").read().strip().splitlines()[-1])
print("config5 50M", round(c["value"]), json.dumps(c.get("config5"))[:900])
PY
