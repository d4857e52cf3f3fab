#!/bin/bash
# what the grid-wide barrier of the one-launch query forward costs by the fences around it (variant libraries built by hand with
# -DFSGPU_Q1L_FENCE=1 / 0; 0 is a timing floor only: without agent-scope fences the stages are not coherent across XCDs)
L=frankensearch_amd/libfsgpu.so
cp $L /tmp/libfsgpu_default.so
for v in 1 0; do
  [ -f frankensearch_amd/libfsgpu_variant_q1l_f$v.so ] || continue
  cp frankensearch_amd/libfsgpu_variant_q1l_f$v.so $L
  echo "one launch, FSGPU_Q1L_FENCE=$v"; SAVE=/tmp/one_f$v.npy timeout 300 python scripts/r06/one_launch_ab.py 2>&1 | grep -v amdgpu.ids
done
cp /tmp/libfsgpu_default.so $L
python - <<'PY'
import numpy as np, os
b = np.load("/tmp/many.npy") if os.path.exists("/tmp/many.npy") else None
for v in (1, 0):
    p = "/tmp/one_f%d.npy" % v
    if b is not None and os.path.exists(p):
        a = np.load(p)
        print("fence", v, "outputs bit-identical to the 25-launch form:", np.array_equal(a.view(np.uint32), b.view(np.uint32)), " max abs diff", float(np.abs(a - b).max()))
PY
