#!/bin/bash
# GPU session 1 of round 4: the scalar-cache repro, the A/B soak of the wide kernel's bitmap loads, the state of the shard / 10M bench
O=gpurun_out/r04s1; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
( time scripts/r04/_build/scache_repro 1000000 ) > $O/scache_repro.txt 2>&1; tail -45 $O/scache_repro.txt
cp frankensearch_amd/libfsgpu.so /tmp/base.so
for v in noinv vecbm; do
  cp frankensearch_amd/libfsgpu_variant_$v.so frankensearch_amd/libfsgpu.so
  python scripts/r04/bitmap_soak.py ${SOAK_SECONDS:-420} 100 > $O/soak_$v.txt 2>&1; echo "== $v"; tail -12 $O/soak_$v.txt
done
cp /tmp/base.so frankensearch_amd/libfsgpu.so
python scripts/r04/bitmap_soak.py ${SOAK_SECONDS:-420} 100 > $O/soak_default.txt 2>&1; echo "== default"; tail -5 $O/soak_default.txt
python bench.py --rows 1250000 --steps 40 --warmup 5 --no-cpu-baseline --no-two-tier --no-adversarial --no-encoders 2>/dev/null | tail -1 > $O/bench_shard.json
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-two-tier --no-adversarial --no-encoders 2>/dev/null | tail -1 > $O/bench_10m.json
python - <<PY
import json
for f in ("bench_shard", "bench_10m"):
    d = json.loads(open("$O/%s.json" % f).read().strip().splitlines()[-1])
    print(f, "qps", round(d["value"]), "ms/step", round(d["ms_per_step"], 4), "main", round(d["roofline"]["avg_launch_ms"], 4), "frac", round(d["roofline"]["frac"], 3))
PY
