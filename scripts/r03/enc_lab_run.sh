#!/bin/bash
# on the GPU box, lab build in place: tolerance tests of the encoder, per-phase stamps of the one-launch forward, the size sweep
export FSGPU_BUILD_DEFS="-DFSGPU_EXPERIMENTS"
python -m pytest tests/test_gpu_bert.py -m gpu -x -q 2>&1 | tail -3
python scripts/r03/enc_stamps.py 2>&1 | grep -A11 "batch 256"
python scripts/r03/enc_sweep.py 2>&1 | grep -E "texts +(8|256|512) "
