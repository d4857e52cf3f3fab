#!/bin/bash
# Lab helper: rebuild ONLY mfma_wide.o of the current (production-flag) build with extra -D switches and relink libfsgpu.so.
# usage: scripts/r03/wide_variant.sh "-DFSGPU_WIDE_EARLY_DMA"
set -e
cd "$(dirname "$0")/../.."
python - "$1" <<'PY'
import os, subprocess, sys
sys.path.insert(0, os.getcwd())
from frankensearch_amd import build as b
extra = sys.argv[1].split()
src = os.path.join(b.CSRC, "mfma_wide.hip")
obj = os.path.join(b.OBJ, "mfma_wide.o")
subprocess.check_call([b._hipcc()] + b.FLAGS + b.EXTRA_FLAGS.get("mfma_wide.hip", []) + b._extra_defs() + extra + ["-I", b.INCLUDE, "-c", src, "-o", obj])
objs = [os.path.join(b.OBJ, os.path.splitext(s)[0] + ".o") for s in b.SOURCES]
subprocess.check_call([b._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", b.LIB] + objs + ["-ldl", "-pthread"])
print("relinked", b.LIB)
PY
