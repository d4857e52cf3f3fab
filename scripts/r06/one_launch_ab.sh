#!/bin/bash
mkdir -p gpurun_out/r06
echo "one launch, grid-wide barriers (default)"; SAVE=/tmp/one.npy timeout 300 python scripts/r06/one_launch_ab.py 2>&1 | grep -v amdgpu.ids
echo "25 launches replayed from a graph (FSGPU_BERT_NO_ONE_LAUNCH=1)"; FSGPU_BERT_NO_ONE_LAUNCH=1 SAVE=/tmp/many.npy timeout 300 python scripts/r06/one_launch_ab.py 2>&1 | grep -v amdgpu.ids
python - <<'PY'
import numpy as np
a, b = np.load("/tmp/one.npy"), np.load("/tmp/many.npy")
print("outputs bit-identical:", np.array_equal(a.view(np.uint32), b.view(np.uint32)), " max abs diff", float(np.abs(a - b).max()))
PY
