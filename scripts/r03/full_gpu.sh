#!/bin/bash
# production build on the GPU box: the whole GPU suite, the encoder fuzzer, encoder micro-benchmarks, a quick default-shape bench line
O=gpurun_out/${1:-r03d}
mkdir -p $O
( time python -m pytest tests -m gpu -q ) > $O/pytest.txt 2>&1; grep -E "passed|failed|real" $O/pytest.txt; grep -E "^E  |^FAILED" $O/pytest.txt | head -20
[ -f tests/fuzz_encoders.py ] && (python tests/fuzz_encoders.py 7 > $O/fuzz_enc.txt 2>&1; tail -2 $O/fuzz_enc.txt)
python scripts/bench_encoders.py > $O/enc.txt 2>&1; grep -E "^(m2v|bert)" $O/enc.txt
python scripts/r03/enc_sweep.py > $O/enc_sweep.txt 2>&1; grep texts $O/enc_sweep.txt
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-adversarial > $O/bench_q.json 2>/dev/null
python - <<PY
import json
d = json.loads(open("$O/bench_q.json").read().strip().splitlines()[-1])
print("qps", round(d["value"]), "e2e", d.get("end_to_end_queries_per_sec"), "enc", json.dumps(d.get("encoders", {}).get("minilm_l6", {}))[:300])
PY
