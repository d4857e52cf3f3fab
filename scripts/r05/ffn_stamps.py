"""Where a block of bert_ffn_w64_kernel spends its life (lab build -DFSGPU_FFN_STAMPS: scripts/r04/build_variant.sh stamps
bert_gemm_w.hip "-DFSGPU_FFN_STAMPS", copied over libfsgpu.so by the caller).  Prints average cycles per phase, waves 0 and 7."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import frankensearch_amd as fa
from frankensearch_amd import _lib
from frankensearch_amd.synthetic import random_bert_weights

rng = np.random.default_rng(0)
bert = fa.NativeEmbedder(random_bert_weights(1, 30522, 384, 6, 1536))
long = [[101] + rng.integers(1000, 30000, 510).tolist() + [102] for _ in range(32)]
offs = np.zeros(33, dtype=np.uint32)
offs[1:] = np.cumsum([len(b) for b in long])
ids = np.concatenate([np.asarray(b, dtype=np.int32) for b in long])
out = np.empty((32, 384), dtype=np.float32)
lib = _lib.lib()
buf = (C.c_ulonglong * 16)()
for _ in range(50):
    bert.embed_flat(ids, offs, out)
lib.fsgpu_lab_ffn_stamps(buf)
for _ in range(20):
    bert.embed_flat(ids, offs, out)
lib.fsgpu_lab_ffn_stamps(buf)
v = np.array(list(buf), dtype=np.float64).reshape(2, 8)
n = v[0, 7]
names = ["phase A (out-proj + LN1)", "up + GELU, half 0", "down, half 0", "up + GELU, half 1", "down, half 1", "epilogue (LN2 + stores)"]
for w, label in ((0, "wave 0"), (1, "wave 7")):
    t = v[w, :7] / n
    print(label, "blocks", int(n), "total cycles %.0f" % (t[6] - t[0]))
    for i, nm in enumerate(names):
        print("   %-28s %8.0f cycles  %5.1f %%" % (nm, t[i + 1] - t[i], 100 * (t[i + 1] - t[i]) / (t[6] - t[0])))
