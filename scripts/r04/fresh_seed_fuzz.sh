#!/bin/bash
# Fresh-seed runs of the three fuzzers on the final build (GPU box).
timeout 200 python scripts/fuzz_batched.py 9101 150 2>&1 | tail -2
timeout 160 python tests/fuzz_exact.py 9103 90 2>&1 | tail -2
timeout 160 python tests/fuzz_encoders.py 9104 90 2>&1 | tail -2
