#!/bin/bash
# Lab helper: rebuild ONLY bert_docs_w.o of an existing lab build (FSGPU_BUILD_DEFS=-DFSGPU_EXPERIMENTS) with extra -D switches and
# relink libfsgpu.so, so that kernel variants can be compared without the three-minute full lab build.
# usage: scripts/r03/enc_variant.sh "-DDOCS_QCH=1 ..."
set -e
cd "$(dirname "$0")/../.."
python - "$1" <<'PY'
import os, subprocess, sys
sys.path.insert(0, os.getcwd())
from frankensearch_amd import build as b
extra = sys.argv[1].split()
src = os.path.join(b.CSRC, "bert_docs_w.hip")
obj = os.path.join(b.OBJ, "bert_docs_w.o")
subprocess.check_call([b._hipcc()] + b.FLAGS + ["-DFSGPU_EXPERIMENTS"] + extra + ["-I", b.INCLUDE, "-c", src, "-o", obj])
objs = [os.path.join(b.OBJ, os.path.splitext(s)[0] + ".o") for s in b.SOURCES]
subprocess.check_call([b._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", b.LIB] + objs + ["-ldl", "-pthread"])
print("relinked", b.LIB)
PY
