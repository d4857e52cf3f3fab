#!/bin/bash
# the adversarial corpora of the default bench line (uniform-random, outlier channels + Zipf clusters) on this build, plus the filter
# tests and the batched fuzzer.   scripts/r05/adversarial.sh OUTDIR
O=${1:-gpurun_out/r05adv}; mkdir -p $O; export TMPDIR=/tmp
( python -m pytest tests/test_gpu_int8_filter.py tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -6 ) | tee $O/pytest.txt
python -m pytest tests/test_gpu_sharded.py -m gpu -q > $O/sharded_alone.log 2>&1; echo "sharded alone rc=$?" | tee -a $O/pytest.txt
python scripts/fuzz_batched.py 606 100 2>&1 | tail -2 | tee $O/fuzz.txt
python - <<'PY' | tee $O/adversarial.txt
import json, sys, torch
sys.path.insert(0, ".")
import bench
dev = torch.device("cuda", 0)
for kind in ("uniform", "outlier"):
    r = bench.adversarial_section(kind, 10_000_000, 384, 10, dev, 0)
    print(json.dumps(r))
PY
