#!/bin/bash
# Regenerates the evidence kept under profiles/<round>/ (run on the GPU box through gpurun; results land in
# gpurun_out/<round>/ and are copied to profiles/<round>/ afterwards).  Counter passes are separate runs with
# --pmc only (never combined with trace domains).
R=${1:-r06}
OUT=gpurun_out/$R
mkdir -p $OUT
export TMPDIR=/tmp
(lscpu | head -20; echo; cat /sys/fs/cgroup/cpu.max 2>/dev/null; nproc) > $OUT/host_cpu.txt 2>&1
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python bench.py --batch 128 --no-cpu-baseline --no-two-tier 2>/dev/null | tail -1 > $OUT/bench_batched_b128.json
python bench.py --batch 1152 --no-cpu-baseline --no-two-tier 2>/dev/null | tail -1 > $OUT/bench_batched_b1152.json
FSGPU_WIDE=0 python bench.py --no-cpu-baseline --no-two-tier 2>/dev/null | tail -1 > $OUT/bench_batched_lds_queries_128.json
python bench.py --exact --batch 1 --no-cpu-baseline --no-two-tier 2>/dev/null | tail -1 > $OUT/bench_exact_b1.json
python bench.py --exact --batch 4 --no-cpu-baseline --no-two-tier 2>/dev/null | tail -1 > $OUT/bench_exact_b4.json
FSGPU_FILTER=f16 python bench.py --no-cpu-baseline --no-two-tier 2>/dev/null | tail -1 > $OUT/bench_batched_f16_filter.json
python bench.py --rows 1000000 --no-cpu-baseline --no-two-tier 2>/dev/null | tail -1 > $OUT/bench_config2_1m.json
python bench.py --rows 1250000 --no-cpu-baseline --no-two-tier 2>/dev/null | tail -1 > $OUT/bench_shard_1m25.json
python bench.py --rows 50000000 --config5 --no-adversarial --no-encoders --no-cpu-baseline --no-two-tier 2>/dev/null | tail -1 > $OUT/bench_config5_50m.json
python bench.py --sharded-handle --gpus 1 --no-config5 2>/dev/null | grep queries_per_sec | tail -1 > $OUT/bench_sharded_handle_1gpu.json
# N > 1 rehearsed on the one GPU: two gloo ranks, the sharded-handle leg over two virtual shards (two-tier + config 5 over sharded handles)
FSGPU_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 10 --warmup 3 --config5-rows 6000000 2>/dev/null | tail -1 > $OUT/bench_rehearsal_gloo_2ranks.json
# round 5: the hybrid layout (query groups x row shards).  Eight virtual shards on the one GPU as 2 x 4 (the default for N = 8) and
# as 1 x 8; four gloo ranks as 2 x 2 through the launcher's path.
python bench.py --sharded-handle --virtual-shards --gpus 8 --no-config5 2>/dev/null | grep queries_per_sec | tail -1 > $OUT/bench_sharded_handle_8virtual_2x4.json
python bench.py --sharded-handle --virtual-shards --gpus 8 --query-groups 1 --no-config5 2>/dev/null | grep queries_per_sec | tail -1 > $OUT/bench_sharded_handle_8virtual_1x8.json
FSGPU_BENCH_BACKEND=gloo python bench.py --gpus 4 --steps 10 --warmup 3 --no-sharded-handle 2>/dev/null | tail -1 > $OUT/bench_rehearsal_gloo_4ranks_2x2.json
FSGPU_BENCH_BACKEND=gloo python bench.py --gpus 8 --steps 5 --warmup 2 --no-sharded-handle 2>/dev/null | tail -1 > $OUT/bench_rehearsal_gloo_8ranks_2x4.json
FSGPU_BENCH_BACKEND=gloo python bench.py --gpus 4 --steps 10 --warmup 3 --strong --no-sharded-handle 2>/dev/null | tail -1 > $OUT/bench_rehearsal_gloo_4ranks_2x2_strong.json
# round 6: the many-queries engine (fshost_two_tier_search_many; dynamic batching of per-query callers), its stages alone, its GPU busy fraction
python scripts/r06/exp_two_tier_many.py 2>&1 | grep "qps=" > $OUT/two_tier_many.txt
python scripts/r06/exp_two_tier_batching.py 2>&1 | grep "threads=" > $OUT/two_tier_batching.txt
python scripts/r06/prof_two_tier_stages.py 2>&1 | grep "ms per" > $OUT/two_tier_stages.txt
CASES=0:1024:0 NQ=16384 rocprofv3 --kernel-trace --output-format csv -d $OUT/many_trace -o many -- python scripts/r06/exp_two_tier_many.py > $OUT/many_trace.log 2>&1
python scripts/r06/gpu_busy.py $(find $OUT/many_trace -name "*kernel_trace.csv" | head -1) 0.25 > $OUT/two_tier_many_gpu_busy.txt 2>&1
rm -rf $OUT/many_trace
scripts/ubench/_build/fp6_skeleton > $OUT/fp6_skeleton.txt 2>&1
python scripts/conformance_minilm.py --selftest > $OUT/conformance_selftest.txt 2>&1
python scripts/r04/filtered_tput.py > $OUT/filtered_tput.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_shard -o bench -- \
    python bench.py --rows 1250000 --no-cpu-baseline --no-two-tier --no-adversarial --no-encoders > $OUT/bench_shard_under_trace.json 2> $OUT/bench_shard_trace.err
# per-kernel time of the default bench command
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- \
    python bench.py --no-cpu-baseline --no-two-tier > $OUT/bench_under_trace.json 2> $OUT/bench_trace.err
# HBM traffic of the same command (FETCH_SIZE / WRITE_SIZE, separate passes)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o bench -- \
    python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-two-tier > $OUT/bench_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o bench -- \
    python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-two-tier > $OUT/bench_pmc_write.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_b1 -o bench -- \
    python bench.py --exact --batch 1 --steps 5 --warmup 2 --no-cpu-baseline --no-two-tier > $OUT/bench_pmc_fetch_b1.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_b4 -o bench -- \
    python bench.py --exact --batch 4 --steps 5 --warmup 2 --no-cpu-baseline --no-two-tier > $OUT/bench_pmc_fetch_b4.log 2>&1
for d in pmc_fetch pmc_write pmc_fetch_b1 pmc_fetch_b4; do python scripts/pmc_summary.py $OUT/$d $OUT/$d.json > /dev/null; done
python - <<PY
import json
out = []
for d in ("pmc_fetch", "pmc_write", "pmc_fetch_b1", "pmc_fetch_b4"):
    for e in json.load(open("$OUT/%s.json" % d)):
        if "fsgpu" in e["kernel"]:
            e["run"] = d
            out.append(e)
json.dump(out, open("$OUT/pmc_summary.json", "w"), indent=1)
PY
# matrix-core / LDS counters of the scan kernels
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -o bench -- \
    python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-two-tier > $OUT/bench_pmc_sq.log 2>&1
python scripts/pmc_summary.py $OUT/pmc_sq $OUT/pmc_sq.json > /dev/null
# encoders
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/enc_trace -o enc -- python scripts/bench_encoders.py > $OUT/enc.log 2>&1
python scripts/bench_encoders.py > $OUT/enc_untraced.log 2>&1
# matrix-pipe utilisation of the encoder kernels (counter pass apart from the trace)
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/enc_pmc -o enc -- python scripts/bench_encoders.py > $OUT/enc_pmc.log 2>&1
python scripts/encoder_pmc_summary.py $OUT/enc_trace $OUT/enc_pmc $OUT/encoder_mfma_pmc.json > $OUT/encoder_mfma_pmc.txt 2>&1
# what one coalesced batch costs through the sharded handle against the unsharded index; fresh-seed fuzzers on this build
PYTHONPATH=. python scripts/r05/batch_overhead.py 2>&1 | grep "nq=" > $OUT/batch_overhead.txt
( timeout 200 python scripts/fuzz_batched.py 9601 150 2>&1 | tail -2; timeout 160 python tests/fuzz_exact.py 9603 90 2>&1 | tail -2; timeout 200 python tests/fuzz_encoders.py 9604 120 2>&1 | tail -2; timeout 200 python scripts/fuzz_sharded.py 9605 150 2>&1 | tail -1 ) > $OUT/fuzz_fresh_seeds.txt 2>&1
# the GPU suite on the same box
( time python -m pytest tests -m gpu -q ) > $OUT/gputest.log 2>&1
ls -R $OUT | head -60
