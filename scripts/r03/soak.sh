#!/bin/bash
# a longer soak of the final build: batched / exact / encoder fuzzers over fresh seeds, the repeated-batch determinism check
O=gpurun_out/soak; mkdir -p $O
for s in 81 82 83 84; do FUZZ_DUMP_DIR=$O python scripts/fuzz_batched.py $s 150 2>&1 | grep -E "filter=|MISMATCH|seed=" | tee -a $O/soak.txt; done
for s in 41 42; do python tests/fuzz_exact.py $s 2>&1 | tail -1 | tee -a $O/soak.txt; done
for s in 51 52; do python tests/fuzz_encoders.py $s 2>&1 | tail -1 | tee -a $O/soak.txt; done
for s in 21 22 23 24 25 26; do python scripts/r03/determinism.py $s 200 2>&1 | grep -E " rep |seed=" | tee -a $O/soak.txt; done
