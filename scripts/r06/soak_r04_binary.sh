#!/bin/bash
# r05 verdict item 8: the r04 BINARY (the tree at 49c50d8: the append path that spilled 20 registers around its hand-written scalar loads),
# built in the configuration that failed on one box in round 4 (ONE glc read of a bitmap word, no per-wave s_dcache_inv:
# -DFSGPU_LAB_NO_DCACHE_INV), as the positive control of the soak on as many boxes as the budget allows; then the shipped library.
#   scripts/r06/soak_r04_binary.sh SECONDS   (the r04 tree lives under scripts/r06/_build/r04tree, built in the authoring container)
T=${1:-200}; mkdir -p gpurun_out/r06; export TMPDIR=/tmp
{
echo "box: $(hostname) unique_id $(cat /sys/class/drm/card*/device/unique_id 2>/dev/null | head -1)"
echo "== r04 binary, single read, no s_dcache_inv"
( cd scripts/r06/_build/r04tree && python scripts/r04/bitmap_soak.py $T 100 2>&1 | grep -v amdgpu.ids | tail -n 6 )
echo "== shipped (r06) library"
python scripts/r04/bitmap_soak.py $((T / 2)) 100 2>&1 | grep -v amdgpu.ids | tail -n 3
} 2>&1 | tee gpurun_out/r06/soak_r04_$(date +%s).txt
