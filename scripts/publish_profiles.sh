#!/bin/bash
# Copies the summaries of scripts/collect_profiles.sh from gpurun_out/<round>/ (scratch) into profiles/<round>/ (tracked).
R=${1:-r06}
SRC=gpurun_out/$R
DST=profiles/$R
mkdir -p $DST
cp $SRC/bench_*.json $SRC/host_cpu.txt $SRC/pmc_summary.json $DST/ 2>/dev/null
rm -f $DST/bench_under_trace.err
cp $SRC/trace/bench_kernel_stats.csv $DST/bench_kernel_stats.csv
cp $SRC/trace/bench_domain_stats.csv $DST/bench_domain_stats.csv
cp $SRC/trace_shard/bench_kernel_stats.csv $DST/bench_shard_1m25_kernel_stats.csv 2>/dev/null
cp $SRC/filtered_tput.txt $DST/filtered_tput.txt 2>/dev/null
cp $SRC/enc_trace/enc_kernel_stats.csv $DST/encoder_kernel_stats.csv
grep -E '^(m2v|bert)' $SRC/enc_untraced.log > $DST/enc_bench.txt
cp $SRC/encoder_mfma_pmc.json $SRC/encoder_mfma_pmc.txt $DST/ 2>/dev/null
cp $SRC/pmc_sq.json $DST/scan_sq_pmc_raw.json
cp $SRC/batch_overhead.txt $SRC/fuzz_fresh_seeds.txt $DST/ 2>/dev/null
cp $SRC/two_tier_many.txt $SRC/two_tier_batching.txt $SRC/two_tier_stages.txt $SRC/two_tier_many_gpu_busy.txt $SRC/fp6_skeleton.txt $SRC/conformance_selftest.txt $DST/ 2>/dev/null
[ -f $SRC/gputest.log ] && grep -E "passed|failed|real" $SRC/gputest.log > $DST/gputest_summary.txt
ls $DST
