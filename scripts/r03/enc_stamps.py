"""Lab build only (FSGPU_BUILD_DEFS=-DFSGPU_EXPERIMENTS, FSGPU_BERT_DOCS_STAMPS=1): per-phase shader-clock stamps of block 0 of the
one-launch MiniLM forward, printed by the library on stderr; this script drives a few calls and prints the median per phase."""
import os, sys, subprocess, re
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np
    import frankensearch_amd as fa
    from frankensearch_amd.synthetic import random_bert_weights
    rng = np.random.default_rng(0)
    bert = fa.NativeEmbedder(random_bert_weights(1, 30522, 384, 6, 1536))
    B = int(sys.argv[2])
    batch = [[101] + rng.integers(1000, 30000, int(rng.integers(6, 31))).tolist() + [102] for _ in range(B)]
    offs = np.zeros(B + 1, dtype=np.uint32); offs[1:] = np.cumsum([len(b) for b in batch])
    flat = np.concatenate([np.asarray(b, dtype=np.int32) for b in batch])
    out = np.empty((B, 384), dtype=np.float32)
    for _ in range(12): bert.embed_flat(flat, offs, out)
    sys.exit(0)
import numpy as np
for B in (8, 256):
    env = dict(os.environ, FSGPU_BERT_DOCS_STAMPS="1")
    res = subprocess.run([sys.executable, os.path.abspath(__file__), "child", str(B)], env=env, capture_output=True, text=True)
    rows = [list(map(int, l.split()[2:])) for l in res.stderr.splitlines() if l.startswith("[docs stamps]")]
    if not rows:
        print("no stamps:", res.stderr[-500:]); continue
    a = np.median(np.array(rows[4:]), axis=0)   # cycles since stamp 0, slots 1..
    names = ["embed"]
    for l in range(6):
        names += [f"L{l} qkv", f"L{l} att", f"L{l} ao", f"L{l} ln1", f"L{l} up", f"L{l} down", f"L{l} ln2"]
    names += ["L5.. (end of loop)", "pool"]
    # slots: 1 = after embed; 2 + 8 l = layer start (after qkv); 3 = after attention; 4 = after AO GEMM; 5 = after LN1; 6 = after up;
    # 7 = after down; 8 = after LN2 (layers 0..4: before the next QKV)
    print(f"batch {B}: total {a[-1]:.0f} cycles")
    prev = 0.0
    for i, v in enumerate(a):
        if v < 0: continue
        print(f"  slot {i + 1:2d}: +{v - prev:8.0f}  (t = {v:8.0f})")
        prev = v
