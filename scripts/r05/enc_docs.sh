#!/bin/bash
# Round 5 encoder checks: tolerance tests + fuzzer slice, the encoder micro-benchmark, and a kernel trace of it (per-kernel times of
# the 32 x 512-token case).   scripts/r05/enc_docs.sh OUTDIR
O=${1:-gpurun_out/r05enc}; mkdir -p $O; export TMPDIR=/tmp
( python -m pytest tests/test_gpu_bert.py -m gpu -q -x 2>&1 | tail -5 ) | tee $O/pytest.txt
python scripts/bench_encoders.py 2>&1 | tee $O/enc_bench.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/enc_trace -o enc -- python scripts/bench_encoders.py > $O/enc_traced.log 2>&1
head -14 $O/enc_trace/*kernel_stats.csv | cut -c1-160 | tee $O/enc_kernels.txt
