#!/bin/bash
# encoder document path (32 x 512 tokens) A/B: shipped build (weight-stationary QKV / FFN-up, split post-attention from 6,144 tokens)
# against variants; per-kernel trace of the shipped build
O=gpurun_out/r04enc; mkdir -p $O; export TMPDIR=/tmp
cp frankensearch_amd/libfsgpu.so /tmp/base.so
for v in base $(ls frankensearch_amd/libfsgpu_variant_*.so 2>/dev/null | sed 's/.*variant_//; s/\.so//'); do
  if [ $v = base ]; then cp /tmp/base.so frankensearch_amd/libfsgpu.so; else cp frankensearch_amd/libfsgpu_variant_$v.so frankensearch_amd/libfsgpu.so; fi
  echo "== $v"; python scripts/bench_encoders.py 2>/dev/null | grep -E "^bert"
done 2>&1 | tee $O/enc_ab.txt
cp /tmp/base.so frankensearch_amd/libfsgpu.so
python -m pytest tests/test_gpu_bert.py -m gpu -q -x 2>&1 | tail -3
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o enc -- python scripts/bench_encoders.py > $O/enc_traced.log 2>&1
head -12 $O/trace/*kernel_stats.csv | cut -c1-170
